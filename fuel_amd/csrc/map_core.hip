// map_core.hip -- map object, occupancy state planes, inflation, queries, host mirrors.
//
// Device layout: the voxel grid is addressed linearly (z fastest) exactly
// like the reference (plan_env/include/plan_env/sdf_map.h:145-147).  Per-voxel predicates that
// the per-cycle kernels consume are kept as BIT-PLANES over that linear address space
// (1 bit/voxel): `occupied` (occ > min_occupancy_log), `unknown` (occ < clamp_min_log-1e-3) and
// `inflated`.  Morphology (inflation, the 6-neighbour frontier test) then becomes shifts of the
// linear bit string, which reproduces the reference's "only 0 <= adr < N is checked" wrap quirk
// (sdf_map.cpp:453-458) for free.
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <mutex>

#include <dirent.h>
#include <unistd.h>

#include <chrono>

#include "fuelmi_internal.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void fuelmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// Hardware queues.  A map owns one stream, a finder two, every busy query slot one; the HIP runtime multiplexes a process's
// streams onto GPU_MAX_HW_QUEUES hardware queues -- FOUR by default -- and two streams that land on one queue time-slice
// (four idle maps in the process took a streaming frame from 0.11 to 0.37 ms; ten optimiser threads: ten solves 2.47 ms
// with 16 queues, 2.65 ms with 4 -- profiles/r05_host_timing.txt).  The runtime latches the variable when it
// initialises (the first HIP call of the process).  Rounds 4-5 set it from a load-time constructor; a dlopen constructor
// that mutates the environment of a possibly multi-threaded host (glibc may move `environ` under a concurrent getenv)
// and silently does nothing when HIP is already up was the wrong place.  Now it is EXPLICIT: fuelmi_init(), called by the
// integrator (the facade's SDFMap::initMap, fuel_amd._lib.lib()) at a point of its choosing, reports whether it could
// still take effect; the constructor only runs when the environment opts in (FUELMI_SET_HW_QUEUES=1).
static int g_hwq_state = FUELMI_HWQ_UNINIT;
static std::mutex g_init_mu;
static bool hsa_runtime_is_up() {  // the ROCr runtime keeps /dev/kfd open from hsa_init() on: HIP initialised <=> it is there
  DIR* d = opendir("/proc/self/fd");
  if (!d) return false;
  bool up = false;
  char path[64], tgt[64];
  while (struct dirent* e = readdir(d)) {
    if (e->d_name[0] == '.') continue;
    snprintf(path, sizeof(path), "/proc/self/fd/%s", e->d_name);
    const ssize_t n = readlink(path, tgt, sizeof(tgt) - 1);
    if (n <= 0) continue;
    tgt[n] = 0;
    if (!strcmp(tgt, "/dev/kfd")) {
      up = true;
      break;
    }
  }
  closedir(d);
  return up;
}
extern "C" int fuelmi_init(int hw_queues) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (g_hwq_state == FUELMI_HWQ_SET || g_hwq_state == FUELMI_HWQ_ENV) return FUELMI_OK;  // (decided earlier; idempotent)
  if (getenv("GPU_MAX_HW_QUEUES")) {  // the environment decides (the robust way: export it in the launch file)
    g_hwq_state = FUELMI_HWQ_ENV;
    return FUELMI_OK;
  }
  if (hsa_runtime_is_up()) {  // too late: the runtime has latched its default of 4
    if (g_hwq_state != FUELMI_HWQ_LATE && !getenv("FUELMI_QUIET"))
      std::fprintf(stderr, "[fuelmi] fuelmi_init: the HIP runtime was initialised before this call and GPU_MAX_HW_QUEUES "
                           "was not in the environment: its streams share the runtime's default of 4 hardware queues "
                           "(export GPU_MAX_HW_QUEUES=16, or call fuelmi_init() before the first HIP call)\n");
    g_hwq_state = FUELMI_HWQ_LATE;
    return FUELMI_OK;
  }
  char buf[16];
  snprintf(buf, sizeof(buf), "%d", hw_queues > 0 ? hw_queues : 16);
  setenv("GPU_MAX_HW_QUEUES", buf, 0 /* keep an existing value */);
  g_hwq_state = FUELMI_HWQ_SET;
  return FUELMI_OK;
}
__attribute__((constructor)) static void fuelmi_default_hw_queues() {
  const char* e = getenv("FUELMI_SET_HW_QUEUES");  // opt-in: hosts that cannot call fuelmi_init() early enough themselves
  if (e && atoi(e) != 0) (void)fuelmi_init(atoi(e) > 1 ? atoi(e) : 16);
}
extern "C" int fuelmi_hw_queues(void) {
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  const int env = e ? atoi(e) : 4;
  return g_hwq_state == FUELMI_HWQ_LATE ? 4 : env;  // (LATE: no value was in the environment when the runtime came up)
}
extern "C" int fuelmi_hw_queues_state(void) { return g_hwq_state; }
// fuelmi_map_create: say it once if the process ended up on the runtime's four queues without anybody having decided so
static void warn_hw_queues_once() {
  static bool said = false;
  if (said || getenv("FUELMI_QUIET")) return;
  if (g_hwq_state == FUELMI_HWQ_UNINIT && !getenv("GPU_MAX_HW_QUEUES")) {
    said = true;
    std::fprintf(stderr, "[fuelmi] GPU_MAX_HW_QUEUES is not set and fuelmi_init() was not called: a map's, its finder's and "
                         "its query threads' streams share the HIP runtime's default of 4 hardware queues\n");
  }
}

// (Round 6 created these streams with hipExtStreamCreateWithCUMask behind an experiment hook -- the finder's streams on
// one, two or four XCDs or on the first 8 / 16 CUs of every XCD, the map's on the rest: no split beat the shared chip
// on any workload, profiles/r06_cu_mask_sweep_*.txt; the hook is gone, the call site stays in one place.)
hipError_t fuelmi_stream_create(hipStream_t* s, int priority, const char* /*which*/) {
  if (priority == INT_MIN) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority);
}

extern "C" const char* fuelmi_last_error(void) { return g_err; }
extern "C" const char* fuelmi_version(void) { return "fuelmi 0.1 (gfx950)"; }
extern "C" int fuelmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}


// ---------------------------------------------------------------------------------------------
// profiling scopes
// ---------------------------------------------------------------------------------------------
StageScope::StageScope(fuelmi_map* m_, int stage_, hipStream_t stream, bool kernel_timed) : m(m_), stage(stage_) {
  st = stream ? stream : m->stream;
  if (!(m->profile_mask & (1u << stage))) return;
  std::lock_guard<std::mutex> lk(m->prof_mu);
  ProfileSlot& s = m->prof[stage];
  if (s.used + 2 > s.ev.size()) {
    size_t old = s.ev.size();
    s.ev.resize(old + 64);
    for (size_t i = old; i < s.ev.size(); ++i) (void)hipEventCreate(&s.ev[i]);
  }
  if (kernel_timed) {
    m->kev[0] = s.ev[s.used], m->kev[1] = s.ev[s.used + 1];
  } else {
    (void)hipEventRecord(s.ev[s.used], st);
    e1 = s.ev[s.used + 1];
  }
  s.used += 2;
}
StageScope::~StageScope() {
  if (e1) (void)hipEventRecord(e1, st);
  if (m->kev[0]) {  // a kernel-timed scope whose launch never happened (error path): keep the pair well-formed
    (void)hipEventRecord(m->kev[0], st);
    (void)hipEventRecord(m->kev[1], st);
    m->kev[0] = m->kev[1] = nullptr;
  }
}

int QuerySlotGuard::acquire(fuelmi_map* m_, size_t bytes) {
  m = m_;
  {
    std::unique_lock<std::mutex> lk(m->qs_mu);
    for (;;) {
      for (auto& q : m->qslots)
        if (!q->busy) {
          s = q.get();
          break;
        }
      if (s) break;
      if (m->qslots.size() < 32) {
        m->qslots.emplace_back(new fuelmi_map::QuerySlot);
        s = m->qslots.back().get();
        break;
      }
      m->qs_cv.wait(lk);
    }
    s->busy = true;
  }
  if (!s->st) {
    HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&s->ev_dep, hipEventDisableTiming));
    // (blocking-sync: finish() polls it with hipEventQuery and, after 3 ms, really BLOCKS in hipEventSynchronize -- without
    // the flag that call spins on ROCm and a long solve kept burning the core)
    HIPCHK(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming | hipEventBlockingSync));
    HIPCHK(hipEventCreateWithFlags(&s->ev_rd, hipEventDisableTiming));
  }
  if (bytes > s->pin_cap) {
    if (s->pin) HIPCHK(hipHostFree(s->pin));
    s->pin = nullptr, s->pin_cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 64 * 1024);
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->pin), want, hipHostMallocDefault));
    s->pin_cap = want;
  }
  // behind everything the map's stream holds now (an ESDF update the caller has just queued, say).  An IDLE map stream
  // holds nothing to wait for: skipping the record + wait then keeps the query off a cross-queue dependency -- with the
  // slot's stream on a hardware queue of its own that dependency alone was 59 of a combineCost's 92 us
  // (profiles/r05_facade_bisect.txt; in round 4 the slot happened to share the map stream's queue)
  m->rw_mu.lock_shared();  // (until the caller's kernel is launched: finish(), or the destructor on an error path)
  reading = true;
  if (hipStreamQuery(m->stream) != hipSuccess) {
    HIPCHK(hipEventRecord(s->ev_dep, m->stream));
    HIPCHK(hipStreamWaitEvent(s->st, s->ev_dep, 0));
  }
  return FUELMI_OK;
}
hipError_t QuerySlotGuard::finish() {
  hipError_t e = hipEventRecord(s->ev_done, s->st);
  if (reading) {  // the query's kernel is on its stream: writers may look at the slots again
    reading = false;
    m->rw_mu.unlock_shared();
  }
  if (e != hipSuccess) return e;
  const bool yld = poll_yields();
  const auto t_begin = std::chrono::steady_clock::now();
  for (long spins = 0;; ++spins) {  // poll: a blocking wait costs ~15 us of wake-up for a 10-us kernel
    e = hipEventQuery(s->ev_done);
    if (e != hipErrorNotReady) return e;
    if (yld) std::this_thread::yield();
    // a query that is not back after ~3 ms of polling (a poll is ~10 ns: a spin COUNT of a few thousand turned a 33-us
    // combineCost into 91 us of poll + blocking wake-up): stop burning the core, block on the event
    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t_begin > std::chrono::milliseconds(3))
      return hipEventSynchronize(s->ev_done);
  }
}
int map_wait_query_readers(fuelmi_map* m) {  // (the caller holds m->rw_mu exclusively)
  std::lock_guard<std::mutex> lk(m->qs_mu);
  for (auto& q : m->qslots)
    if (q->busy && q->st && q->ev_rd) {
      HIPCHK(hipEventRecord(q->ev_rd, q->st));
      HIPCHK(hipStreamWaitEvent(m->stream, q->ev_rd, 0));
    }
  return FUELMI_OK;
}
QuerySlotGuard::~QuerySlotGuard() {
  if (reading) {
    reading = false;
    m->rw_mu.unlock_shared();
  }
  if (!s) return;
  (void)hipStreamSynchronize(s->st);  // (an error path may leave work behind: the pinned block is about to be reused)
  {
    std::lock_guard<std::mutex> lk(m->qs_mu);
    s->busy = false;
  }
  m->qs_cv.notify_one();
}

int map_ensure_stage(fuelmi_map* m, size_t dev_bytes, size_t host_bytes) {
  if (dev_bytes > m->d_stage_bytes) {
    if (m->d_stage) HIPCHK(hipFree(m->d_stage));
    m->d_stage = nullptr;
    m->d_stage_bytes = 0;
    size_t want = dev_bytes + dev_bytes / 4 + 4096;
    HIPCHK(hipMalloc(&m->d_stage, want));
    m->d_stage_bytes = want;
  }
  if (host_bytes > m->h_stage_bytes) {
    if (m->h_stage) HIPCHK(hipHostFree(m->h_stage));
    m->h_stage = nullptr;
    m->h_stage_bytes = 0;
    size_t want = host_bytes + host_bytes / 4 + 4096;
    HIPCHK(hipHostMalloc(&m->h_stage, want, hipHostMallocDefault));
    m->h_stage_bytes = want;
  }
  return FUELMI_OK;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// occupancy log-odds -> (occupied, unknown) bit-planes; one lane per voxel, one ballot per word.
// thresholds exactly as SDFMap::getOccupancy (sdf_map.h:196-203) in f64.
__global__ void __launch_bounds__(256)
k_state_planes(Geo g, const double* __restrict__ occ, u64* __restrict__ occ_bits,
               u64* __restrict__ unk_bits, double thr_occ, double thr_unk, int w_lo, int w_hi) {
  long a = 64L * w_lo + (long)blockIdx.x * blockDim.x + threadIdx.x;
  long aend = 64L * (w_hi + 1);
  for (; a < aend; a += (long)gridDim.x * blockDim.x) {
    bool in = a < g.N;
    double o = in ? occ[a] : 0.0;
    u64 mo = __ballot(in && o > thr_occ);
    u64 mu = __ballot(in && o < thr_unk);
    if ((threadIdx.x & 63) == 0) {
      occ_bits[a >> 6] = mo;
      unk_bits[a >> 6] = mu;
    }
  }
}

__global__ void k_fill_f64(double* p, double v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill_f32(float* p, float v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// S[w] = occupied[w] & box(w) for w in [w_lo, w_hi] (may extend into the zero margins)
__global__ void __launch_bounds__(256)
k_box_and(Geo g, Box3 b, const u64* __restrict__ src, u64* __restrict__ dst, int w_lo, int w_hi) {
  int w = w_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (w > w_hi) return;
  u64 v = 0ull;
  if (w >= 0 && w < g.W) v = src[w] & box_mask_word(g, w, b);
  dst[w] = v;
}

// SDFMap::clearAndInflateLocalMap (sdf_map.cpp:434-462) on bit-planes:
//   infl = (infl & ~box) | OR_{dx,dy,dz in [-s,s]} shift(S, dx*ny*nz + dy*nz + dz), clipped to [0,N)
// The stamp is a cube, and shifts of the linear bit string compose, so the (2s+1)^3 shifted copies
// factor into a y/z dilation followed by an x dilation (5 + 5 window pairs per word instead of 25;
// the intermediate plane keeps the bits that leave [0,N) in its margins, so the reference's
// "only the final address is range-checked" wrap quirk is preserved exactly).
// One thread per output word; the dz loop is a funnel-shifted OR over a 128-bit window.
// STEP > 0: the step is a compile-time constant (2 for the usual 0.199 m inflation at 0.1 m) -- all plane words of
// the 2*STEP+1 windows are fetched before the first is used (one memory round trip instead of one per dy: the kernel
// is a chain of dependent L2 reads otherwise); STEP == 0: any step, the loop form.
template <int STEP>
__global__ void __launch_bounds__(256)
k_inflate_yz(Geo g, int step, const u64* __restrict__ S, u64* __restrict__ T, int w_lo, int w_hi) {
  int w = w_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (w > w_hi) return;
  u64 acc = 0ull;
  if (STEP > 0) {
    u64 a[2 * STEP + 1][3];
#pragma unroll
    for (int i = 0; i < 2 * STEP + 1; ++i) {
      const long start = 64L * w - (long)(i - STEP) * g.nz - STEP;
      const long wi = start >> 6;
      a[i][0] = S[wi], a[i][1] = S[wi + 1], a[i][2] = S[wi + 2];
    }
#pragma unroll
    for (int i = 0; i < 2 * STEP + 1; ++i) {
      const long start = 64L * w - (long)(i - STEP) * g.nz - STEP;
      const int sh = (int)(start & 63);
      const u64 lo = sh ? (a[i][0] >> sh) | (a[i][1] << (64 - sh)) : a[i][0];
      const u64 hi = sh ? (a[i][1] >> sh) | (a[i][2] << (64 - sh)) : a[i][1];
      u64 r = lo;
#pragma unroll
      for (int k = 1; k <= 2 * STEP; ++k) r |= (lo >> k) | (hi << (64 - k));
      acc |= r;
    }
  } else {
    for (int dy = -step; dy <= step; ++dy) {
      long start = 64L * w - (long)dy * g.nz - step;
      u64 lo = plane_window(S, start);
      u64 hi = plane_window(S, start + 64);
      if ((lo | hi) == 0ull) continue;
      u64 r = lo;
      for (int k = 1; k <= 2 * step; ++k) r |= (lo >> k) | (hi << (64 - k));
      acc |= r;
    }
  }
  T[w] = acc;
}
__global__ void __launch_bounds__(256)
k_inflate_x(Geo g, Box3 b, int step, const u64* __restrict__ T, u64* __restrict__ infl, int w_lo, int w_hi) {
  // every word is read by the 2*step+1 workgroups that hold its x-neighbours: give each XCD (workgroup i
  // runs on XCD i % 8, own L2) one contiguous eighth of the word range instead of every eighth block
  const int per_xcd = (gridDim.x + 7) >> 3;
  const int lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  int w = w_lo + lb * blockDim.x + threadIdx.x;
  if (w > w_hi) return;
  u64 acc = 0ull;
  for (int dx = -step; dx <= step; ++dx) acc |= plane_window(T, 64L * w - (long)dx * g.nyz);
  long a0 = 64L * w;
  if (a0 + 64 > g.N) acc &= bit_range(0, (int)(g.N - a0));
  infl[w] = (infl[w] & ~box_mask_word(g, w, b)) | acc;
}

// The same inflation in ONE launch (round 4).  The two kernels above move 10 MB and take 12 us on the 400^2 x 100 map:
// two launches, a boundary and a plane written and read back in between, each a chain of dependent window loads.
// Here a workgroup owns 256 output words; for each of the 2*STEP+1 x offsets it stages the source words its outputs
// can reach (256 + the y/z reach), box-masked, in LDS, dilates them in z there (shifts of the linear bit string across
// word boundaries), and every lane then ORs the (2*STEP+1)^2 (dx, dy) windows of its word out of LDS.  Same bits as
// the factored form: out[a] = OR S[a - dx*ny*nz - dy*nz + dz], S = occupied & box on the linear address string, only
// the final address range-checked (the reference's wrap quirk, sdf_map.cpp:453-458).
template <int STEP, bool WHOLE>
__global__ void __launch_bounds__(256)
k_inflate_fused(Geo g, Box3 b, const u64* __restrict__ occ_bits, u64* __restrict__ infl, int w_lo, int w_hi, int margin) {
  constexpr int NR = 2 * STEP + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ry = (STEP * (g.nz + 1) + 63) / 64 + 1;  // y/z reach of an output word, in words
  const int RW = 256 + 2 * ry + 2;                   // words staged per x offset (+ the funnels' second word)
  u64* raw = reinterpret_cast<u64*>(smem_raw);       // [NR][RW + 2]  (one more word either side for the z dilation)
  u64* sz = raw + (size_t)NR * (RW + 2);             // [NR][RW]
  const int w0 = w_lo + blockIdx.x * 256;
  for (int it = threadIdx.x; it < NR * (RW + 2); it += 256) {
    const int r = it / (RW + 2), k = it - r * (RW + 2);
    const long start = 64L * w0 - (long)(r - STEP) * g.nyz;
    const long wi = (start >> 6) - ry - 1 + k;
    u64 v = 0ull;
    if (wi >= -margin && wi < (long)g.W + margin) {
      v = occ_bits[wi];
      if (!WHOLE) v = (wi >= 0 && wi < g.W) ? (v & box_mask_word(g, (int)wi, b)) : 0ull;
    }
    raw[it] = v;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < NR * RW; it += 256) {
    const int r = it / RW, k = it - r * RW;
    const u64* q = raw + (size_t)r * (RW + 2) + k;  // q[0], q[1], q[2] = words wi - 1, wi, wi + 1 of range word k
    const u64 s0 = q[0], s1 = q[1], s2 = q[2];
    u64 zd = s1;
#pragma unroll
    for (int d = 1; d <= STEP; ++d) zd |= (s1 >> d) | (s2 << (64 - d)) | (s1 << d) | (s0 >> (64 - d));
    sz[it] = zd;
  }
  __syncthreads();
  const int w = w0 + threadIdx.x;
  if (w > w_hi) return;
  u64 acc = 0ull;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const long start = 64L * w0 - (long)(r - STEP) * g.nyz;
    const int shx = (int)(start & 63);
    const u64* row = sz + (size_t)r * RW;
#pragma unroll
    for (int dy = -STEP; dy <= STEP; ++dy) {
      const int rel = 64 * ((int)threadIdx.x + ry) + shx - dy * g.nz;  // bit position relative to the staged range
      const int kw = rel >> 6, sh = rel & 63;
      const u64 lo = row[kw], hi = row[kw + 1];
      acc |= sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    }
  }
  const long a0 = 64L * w;
  if (a0 + 64 > g.N) acc &= bit_range(0, (int)(g.N - a0));
  infl[w] = WHOLE ? acc : ((infl[w] & ~box_mask_word(g, w, b)) | acc);
}

// virtual ceiling (sdf_map.cpp:464-471): occupancy_buffer_[x,y,ceil_id] = clamp_max_log
__global__ void k_virtual_ceil(Geo g, Box3 b, int ceil_id, double lmax, double* occ, u64* occ_bits,
                               u64* unk_bits, double thr_occ, double thr_unk) {
  int nxb = b.hi[0] - b.lo[0] + 1, nyb = b.hi[1] - b.lo[1] + 1;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nxb * nyb) return;
  int x = b.lo[0] + i / nyb, y = b.lo[1] + i % nyb;
  long a = (long)x * g.nyz + (long)y * g.nz + ceil_id;
  occ[a] = lmax;
  u64 bit = 1ull << (a & 63);
  if (lmax > thr_occ) atomicOr(&occ_bits[a >> 6], bit);
  else atomicAnd(&occ_bits[a >> 6], ~bit);
  if (lmax < thr_unk) atomicOr(&unk_bits[a >> 6], bit);
  else atomicAnd(&unk_bits[a >> 6], ~bit);
}

// SDFMap::setOccupied (sdf_map.h:210-215)
__global__ void k_set_occupied(Geo g, const double* __restrict__ pos, int n, int occv, u64* infl) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  for (int k = 0; k < 3; ++k)
    if (p[k] < g.minb[k] + 1e-4 || p[k] > g.maxb[k] - 1e-4) return;
  int id[3];
  for (int k = 0; k < 3; ++k) id[k] = (int)floor((p[k] - g.org[k]) * g.res_inv);
  long a = (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2];
  if (a < 0 || a >= g.N) return;
  u64 bit = 1ull << (a & 63);
  if (occv)
    atomicOr(&infl[a >> 6], bit);
  else
    atomicAnd(&infl[a >> 6], ~bit);
}

// SDFMap::resetBuffer(min,max) (sdf_map.cpp:101-114): inflate = 0, distance = default in the box
__global__ void k_reset_bits(Geo g, Box3 b, u64* infl, int w_lo, int w_hi) {
  int w = w_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (w > w_hi) return;
  infl[w] &= ~box_mask_word(g, w, b);
}
__global__ void k_reset_dist(Geo g, Box3 b, float* dist, float v) {
  // grid: x = z-line chunks; one thread per voxel of the box
  int zl = b.hi[2] - b.lo[2] + 1, yl = b.hi[1] - b.lo[1] + 1, xl = b.hi[0] - b.lo[0] + 1;
  long n = (long)xl * yl * zl;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) {
    int z = (int)(i % zl);
    long r = i / zl;
    int y = (int)(r % yl), x = (int)(r / yl);
    dist[(long)(b.lo[0] + x) * g.nyz + (long)(b.lo[1] + y) * g.nz + b.lo[2] + z] = v;
  }
}

// bits -> bytes / f32 -> f64 expansion for the host mirrors (contiguous address range)
__global__ void k_expand_bits(const u64* __restrict__ bits, long a0, long n, char* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) {
    long a = a0 + i;
    out[i] = (char)((bits[a >> 6] >> (a & 63)) & 1ull);
  }
}
__global__ void k_expand_dist(const float* __restrict__ dist, long a0, long n, double res,
                              double* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) out[i] = dist_to_f64(dist[a0 + i], res);
}

// Mirror refresh of an index box (z fastest: a wave writes whole z-row pieces).  DIRECT: the outputs are the
// caller's full-size host buffers mapped into the device address space (posted PCIe writes of exactly the box
// bytes); otherwise compact [x][y][z] staging arrays.
template <bool DIRECT>
__global__ void __launch_bounds__(256)
k_sync_box(Geo g, Box3 b, const double* __restrict__ occ, const u64* __restrict__ infl, const float* __restrict__ dist,
           double* __restrict__ o_occ, char* __restrict__ o_infl, double* __restrict__ o_dist) {
  const int zlen = b.hi[2] - b.lo[2] + 1, ylen = b.hi[1] - b.lo[1] + 1, xlen = b.hi[0] - b.lo[0] + 1;
  const long total = (long)xlen * ylen * zlen;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long line = i / zlen;
    const int zi = (int)(i - line * zlen);
    const int xi = (int)(line / ylen), yi = (int)(line - (long)xi * ylen);
    const long a = (long)(b.lo[0] + xi) * g.nyz + (long)(b.lo[1] + yi) * g.nz + b.lo[2] + zi;
    const long o = DIRECT ? a : i;
    if (o_occ) o_occ[o] = occ[a];
    if (o_infl) o_infl[o] = (char)((infl[a >> 6] >> (a & 63)) & 1ull);
    if (o_dist) o_dist[o] = dist_to_f64(dist[a], g.res);
  }
}

__global__ void k_dist_grad(Geo g, const float* __restrict__ dist, const double* __restrict__ pos, int n,
                            double* __restrict__ out_d, double* __restrict__ out_g) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, gr[3];
  out_d[i] = dist_with_grad_dev(g, dist, p, gr);
  out_g[3 * i] = gr[0];
  out_g[3 * i + 1] = gr[1];
  out_g[3 * i + 2] = gr[2];
}
__global__ void k_coarse_dist(Geo g, const float* __restrict__ dist, const double* __restrict__ pos, int n,
                              double* __restrict__ out_d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int id[3];
  for (int k = 0; k < 3; ++k) id[k] = (int)floor((pos[3 * i + k] - g.org[k]) * g.res_inv);
  out_d[i] = get_distance_idx(g, dist, id[0], id[1], id[2]);
}
__global__ void k_query_state(Geo g, const u64* __restrict__ occ_bits, const u64* __restrict__ unk_bits,
                              const u64* __restrict__ infl_bits, const int* __restrict__ idx, int n,
                              int* __restrict__ o_occ, int* __restrict__ o_infl) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = idx[3 * i], y = idx[3 * i + 1], z = idx[3 * i + 2];
  if (x < 0 || y < 0 || z < 0 || x > g.nx - 1 || y > g.ny - 1 || z > g.nz - 1) {
    o_occ[i] = -1;
    o_infl[i] = -1;
    return;
  }
  long a = (long)x * g.nyz + (long)y * g.nz + z;
  u64 bit = 1ull << (a & 63);
  int st = (unk_bits[a >> 6] & bit) ? 0 : ((occ_bits[a >> 6] & bit) ? 2 : 1);
  o_occ[i] = st;
  o_infl[i] = (infl_bits[a >> 6] & bit) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// host API
// ---------------------------------------------------------------------------------------------
int plane_alloc(fuelmi_map* m, Plane& pl) {
  size_t words = (size_t)m->g.W + 2 * (size_t)m->margin_words + 2;
  HIPCHK(hipMalloc(&pl.base, words * sizeof(u64)));
  HIPCHK(hipMemsetAsync(pl.base, 0, words * sizeof(u64), m->stream));
  pl.p = pl.base + m->margin_words;
  return FUELMI_OK;
}

static inline void pos_to_index(const fuelmi_map* m, const double p[3], int id[3]) {
  for (int i = 0; i < 3; ++i) id[i] = (int)std::floor((p[i] - m->g.org[i]) * m->g.res_inv);
}
static inline void bound_index(const fuelmi_map* m, int id[3]) {
  const int nv[3] = {m->g.nx, m->g.ny, m->g.nz};
  for (int i = 0; i < 3; ++i) id[i] = std::max(std::min(id[i], nv[i] - 1), 0);
}
static inline long adr_of(const Geo& g, const int id[3]) {
  return (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2];
}
static inline int blocks_for(long n, int threads, int cap = 1 << 20) {
  long b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

extern "C" int fuelmi_map_create(const fuelmi_map_cfg* c, fuelmi_map** out) {
  ARGCHK(c && out);
  ARGCHK(c->resolution > 0 && c->map_size[0] > 0 && c->map_size[1] > 0 && c->map_size[2] > 0);
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fuelmi_set_error("no HIP device available: libfuelmi has no CPU fallback");
    return FUELMI_ENODEV;
  }
  ARGCHK(c->device >= 0 && c->device < ndev);
  HIPCHK(hipSetDevice(c->device));
  warn_hw_queues_once();
  fuelmi_map* m = new fuelmi_map;
  m->cfg = *c;
  m->device = c->device;
  // --- derived constants exactly as SDFMap::initMap (sdf_map.cpp:30-56,78-84) ---
  fuelmi_map_info& I = m->info;
  Geo& g = m->g;
  g.res = c->resolution;
  g.res_inv = 1 / c->resolution;
  I.resolution_inv = g.res_inv;
  g.org[0] = -c->map_size[0] / 2.0;
  g.org[1] = -c->map_size[1] / 2.0;
  g.org[2] = c->ground_height;
  int nv[3];
  for (int i = 0; i < 3; ++i) {
    nv[i] = (int)std::ceil(c->map_size[i] / c->resolution);
    I.voxel_num[i] = nv[i];
    I.origin[i] = g.org[i];
    g.minb[i] = I.min_boundary[i] = g.org[i];
    g.maxb[i] = I.max_boundary[i] = g.org[i] + c->map_size[i];
  }
  long N = (long)nv[0] * nv[1] * nv[2];
  if (N <= 0 || N >= (1L << 31) - 64) {
    fuelmi_set_error("grid of %ld voxels exceeds the 2^31 address limit", N);
    delete m;
    return FUELMI_ELIMIT;
  }
  if (nv[2] > 255) {
    fuelmi_set_error("nz=%d > 255 unsupported (ESDF z-pass tile holds dz^2 in 16 bits)", nv[2]);
    delete m;
    return FUELMI_ELIMIT;
  }
  g.nx = nv[0], g.ny = nv[1], g.nz = nv[2];
  g.nyz = nv[1] * nv[2];
  g.N = (int)N;
  g.W = (int)((N + 63) / 64);
  auto logit = [](double x) { return std::log(x / (1 - x)); };
  I.prob_hit_log = logit(c->p_hit);
  I.prob_miss_log = logit(c->p_miss);
  I.clamp_min_log = logit(c->p_min);
  I.clamp_max_log = logit(c->p_max);
  I.min_occupancy_log = logit(c->p_occ);
  I.inflate_step = (int)std::ceil(c->obstacles_inflation / c->resolution);
  if (I.inflate_step < 0 || I.inflate_step > 31) {  // the z dilation shifts a 64-bit window by up to `step` bits
    fuelmi_set_error("obstacles_inflation / resolution = %d voxels: the inflation stamp is limited to 31", I.inflate_step);
    delete m;
    return FUELMI_ELIMIT;
  }
  pos_to_index(m, c->box_min, I.box_min);
  pos_to_index(m, c->box_max, I.box_max);
  for (int i = 0; i < 3; ++i) m->local_bound.lo[i] = m->local_bound.hi[i] = 0;
  for (int i = 0; i < 3; ++i) m->upd_min[i] = m->upd_max[i] = 0.0;
  m->reset_updated_box = true;
  int step = std::max(I.inflate_step, 1);
  m->margin_words = (int)(((long)step * ((long)g.nyz + g.nz + 1) + 64) / 64 + 4);

  int rc = FUELMI_OK;
  auto fail = [&](int code) {
    fuelmi_map_destroy(m);
    return code;
  };
  if (fuelmi_stream_create(&m->stream, INT_MIN, "MAP") != hipSuccess) {
    fuelmi_set_error("hipStreamCreate failed");
    return fail(FUELMI_EHIP);
  }
  if ((rc = plane_alloc(m, m->occ_bits)) || (rc = plane_alloc(m, m->unk_bits)) ||
      (rc = plane_alloc(m, m->infl_bits)) || (rc = plane_alloc(m, m->tmp_bits)) || (rc = plane_alloc(m, m->tmp2_bits)) ||
      (rc = plane_alloc(m, m->hit_bits)) || (rc = plane_alloc(m, m->miss_bits)))
    return fail(rc);
  size_t Npad = (size_t)g.W * 64;
  if (hipMalloc(&m->occ, Npad * sizeof(double)) != hipSuccess ||
      hipMalloc(&m->dist, Npad * sizeof(float)) != hipSuccess ||
      hipMalloc(&m->esdf_tmp, Npad * sizeof(u32)) != hipSuccess ||
      hipMalloc(&m->flag_rayend, Npad) != hipSuccess ||
      hipMalloc(&m->ray_owner, Npad * sizeof(u32)) != hipSuccess) {
    fuelmi_set_error("hipMalloc of %zu-voxel grid failed", Npad);
    return fail(FUELMI_ENOMEM);
  }
  if ((g.nz % 4) == 0) {  // the packed ESDF family's 16-bit hand-over (esdf.hip): column tiles of 32 x x-pairs x 128 B
    const size_t ntiles = ((size_t)g.ny * (size_t)((g.nz + 7) / 4 + 1) + 7) / 8 + PK2_MAXCH + 1;  // (8 segments per tile, ragged chunks)
    m->esdf_tmp16_bytes = ntiles * (size_t)(((g.nx + 1) / 2 + 63) / 64) * 64 * 128;  // (tile rows of 128 B per x-pair)
    // (+ ~2.3 B/voxel beside the 4 B/voxel of esdf_tmp: 300 MB on the 800^2 x 200 map.  Not fatal when it does not fit: the
    // packed family then reports ESDF_NO_FIT and the 32-bit family runs -- ADVICE r5)
    if (hipMalloc(reinterpret_cast<void**>(&m->esdf_tmp16), m->esdf_tmp16_bytes) != hipSuccess) {
      (void)hipGetLastError();
      m->esdf_tmp16 = nullptr;
      m->esdf_tmp16_bytes = 0;
    }
  }
  // initial state (sdf_map.cpp:61-72): all unknown, inflate 0, distance default, flag_rayend -1
  k_fill_f64<<<blocks_for(Npad, 256, 65536), 256, 0, m->stream>>>(m->occ, I.clamp_min_log - 0.01, (long)Npad);
  k_fill_f32<<<blocks_for(Npad, 256, 65536), 256, 0, m->stream>>>(m->dist, (float)c->default_dist, (long)Npad);
  (void)hipMemsetAsync(m->esdf_tmp, 0, Npad * sizeof(u32), m->stream);
  (void)hipMemsetAsync(m->flag_rayend, 0xFF, Npad, m->stream);
  (void)hipMemsetAsync(m->ray_owner, 0xFF, Npad * sizeof(u32), m->stream);
  k_state_planes<<<blocks_for(Npad, 256, 65536), 256, 0, m->stream>>>(
      g, m->occ, m->occ_bits.p, m->unk_bits.p, I.min_occupancy_log, I.clamp_min_log - 1e-3, 0, g.W - 1);
  if (hipMalloc(reinterpret_cast<void**>(&m->ins_head), 16 * sizeof(u64)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&m->h_ins), 16 * sizeof(u64), hipHostMallocDefault) != hipSuccess) {
    fuelmi_set_error("allocation of the fusion's result words failed");
    return fail(FUELMI_ENOMEM);
  }
  (void)hipMemsetAsync(m->ins_head, 0, 16 * sizeof(u64), m->stream);
  if (hipMalloc(reinterpret_cast<void**>(&m->esdf_stat), (2 * 256 + 4) * sizeof(u32)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&m->h_esdf_stat), (2 * 256 + 4) * sizeof(u32), hipHostMallocDefault) != hipSuccess) {
    fuelmi_set_error("allocation of the ESDF statistic failed");
    return fail(FUELMI_ENOMEM);
  }
  (void)hipMemsetAsync(m->esdf_stat, 0, (2 * 256 + 4) * sizeof(u32), m->stream);
  memset(m->h_esdf_stat, 0, (2 * 256 + 4) * sizeof(u32));
  memset(m->h_ins, 0, 16 * sizeof(u64));
  if (hipEventCreate(&m->t0) != hipSuccess || hipEventCreate(&m->t1) != hipSuccess || hipEventCreate(&m->t_prof0) != hipSuccess ||
      hipEventCreateWithFlags(&m->ev_planes, hipEventDisableTiming) != hipSuccess ||
      hipStreamSynchronize(m->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    fuelmi_set_error("device initialisation failed (is this a gfx950 device?)");
    return fail(FUELMI_EHIP);
  }
  *out = m;
  return FUELMI_OK;
}

extern "C" void fuelmi_map_destroy(fuelmi_map* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  {
    std::lock_guard<std::mutex> lk(m->dep_mu);
    for (auto& d : m->dependents) d.orphan(d.obj);  // finders / batches that outlive their map
    m->dependents.clear();
  }
  for (auto& r : m->mirror)
    if (r.host) (void)hipHostUnregister(r.host);
  Plane* planes[] = {&m->occ_bits, &m->unk_bits,  &m->infl_bits, &m->tmp_bits,
                     &m->tmp2_bits, &m->hit_bits, &m->miss_bits};
  for (Plane* p : planes)
    if (p->base) (void)hipFree(p->base);
  void* bufs[] = {m->occ, m->dist, m->esdf_tmp, m->esdf_tmp16, m->flag_rayend, m->ray_owner, m->d_stage, m->ins_partial, m->ins_head, m->ins_rec};
  if (m->h_ins) (void)hipHostFree(m->h_ins);
  for (auto& q : m->qslots) {
    if (q->st) (void)hipStreamSynchronize(q->st), (void)hipStreamDestroy(q->st);
    if (q->ev_dep) (void)hipEventDestroy(q->ev_dep);
    if (q->ev_done) (void)hipEventDestroy(q->ev_done);
    if (q->ev_rd) (void)hipEventDestroy(q->ev_rd);
    if (q->pin) (void)hipHostFree(q->pin);
  }
  m->qslots.clear();
  if (m->h_esdf_stat) (void)hipHostFree(m->h_esdf_stat);
  if (m->esdf_stat) (void)hipFree(m->esdf_stat);
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (m->h_stage) (void)hipHostFree(m->h_stage);
  for (auto& s : m->prof)
    for (auto e : s.ev) (void)hipEventDestroy(e);
  if (m->t0) (void)hipEventDestroy(m->t0);
  if (m->t_prof0) (void)hipEventDestroy(m->t_prof0);
  if (m->t1) (void)hipEventDestroy(m->t1);
  if (m->ev_planes) (void)hipEventDestroy(m->ev_planes);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

extern "C" int fuelmi_map_get_info(const fuelmi_map* m, fuelmi_map_info* info) {
  ARGCHK(m && info);
  *info = m->info;
  return FUELMI_OK;
}

extern "C" int fuelmi_map_upload_occupancy(fuelmi_map* m, const double* occ) {
  if (m) ++m->occ_epoch;
  ARGCHK(m && occ);
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  HIPCHK(map_wait_plane_readers(m));
  ++m->fusion_count;
  HIPCHK(hipMemcpyAsync(m->occ, occ, (size_t)g.N * sizeof(double), hipMemcpyHostToDevice, m->stream));
  k_state_planes<<<blocks_for((long)g.W * 64, 256, 65536), 256, 0, m->stream>>>(
      g, m->occ, m->occ_bits.p, m->unk_bits.p, m->info.min_occupancy_log, m->info.clamp_min_log - 1e-3, 0,
      g.W - 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(m->ev_planes, m->stream));
  ++m->planes_ver;
  HIPCHK(hipStreamSynchronize(m->stream));
  return FUELMI_OK;
}

extern "C" int fuelmi_map_inflate_local(fuelmi_map* m) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  const Box3& b = m->local_bound;
  const int step = m->info.inflate_step;
  long reach = (long)step * ((long)g.nyz + g.nz + 1);
  long a_lo = adr_of(g, b.lo) - reach, a_hi = adr_of(g, b.hi) + reach;
  int out_lo = (int)std::max(0L, a_lo >> 6), out_hi = (int)std::min((long)g.W - 1, a_hi >> 6);
  // T (y/z-dilated sources) must be valid wherever the x pass reads, S wherever the y/z pass reads
  const int rx = (int)(((long)step * g.nyz + 127) / 64) + 1;
  const int ryz = (int)(((long)step * (g.nz + 1) + step + 127) / 64) + 1;
  const int t_lo = std::max(out_lo - rx, -m->margin_words), t_hi = std::min(out_hi + rx, g.W - 1 + m->margin_words);
  const int s_lo = std::max(t_lo - ryz, -m->margin_words), s_hi = std::min(t_hi + ryz, g.W - 1 + m->margin_words);
  {
    StageScope sc(m, FUELMI_K_INFLATE);
    // sources = occupied voxels inside the box.  A box that is the whole map needs no masking: the occupancy
    // plane itself (zero margins included) is the source plane
    const bool whole = b.lo[0] == 0 && b.lo[1] == 0 && b.lo[2] == 0 && b.hi[0] == g.nx - 1 && b.hi[1] == g.ny - 1 &&
                       b.hi[2] == g.nz - 1;
    // The fused kernel re-reads every source word five times: free while the words it touches sit in the caches (400^2 x
    // 100: 9.8 against 11.8 us and a launch saved; the local box of a streaming frame on any map), not for the full box of
    // the HBM-resident 800^2 x 200 map (53.6 against 42 us): by the size of the BOX's address range, like every other
    // kernel choice.
    const bool two_pass = (long)(out_hi - out_lo + 1) * 64 > (48L << 20);
    const int ry = (step * (g.nz + 1) + 63) / 64 + 1;
    const size_t lds_f = (size_t)(2 * step + 1) * (2 * (size_t)(256 + 2 * ry + 2) + 2) * sizeof(u64);
    m->last_inflate_kernel = (step == 2 && !two_pass && lds_f <= 64 * 1024) ? 0 : 1;
    if (step == 2 && !two_pass && lds_f <= 64 * 1024) {
      const int nb = blocks_for(out_hi - out_lo + 1, 256);
      if (whole)
        k_inflate_fused<2, true><<<nb, 256, lds_f, m->stream>>>(g, b, m->occ_bits.p, m->infl_bits.p, out_lo, out_hi, m->margin_words);
      else
        k_inflate_fused<2, false><<<nb, 256, lds_f, m->stream>>>(g, b, m->occ_bits.p, m->infl_bits.p, out_lo, out_hi, m->margin_words);
    } else {
    const u64* S = whole ? m->occ_bits.p : m->tmp_bits.p;
    if (!whole)
      k_box_and<<<blocks_for(s_hi - s_lo + 1, 256), 256, 0, m->stream>>>(g, b, m->occ_bits.p, m->tmp_bits.p, s_lo, s_hi);
    if (step == 2)
      k_inflate_yz<2><<<blocks_for(t_hi - t_lo + 1, 256), 256, 0, m->stream>>>(g, step, S, m->tmp2_bits.p, t_lo, t_hi);
    else
      k_inflate_yz<0><<<blocks_for(t_hi - t_lo + 1, 256), 256, 0, m->stream>>>(g, step, S, m->tmp2_bits.p, t_lo, t_hi);
    k_inflate_x<<<((blocks_for(out_hi - out_lo + 1, 256) + 7) / 8) * 8, 256, 0, m->stream>>>(g, b, step, m->tmp2_bits.p,
                                                                             m->infl_bits.p, out_lo, out_hi);
    }
  }
  if (m->cfg.virtual_ceil_height > -0.5) {
    int ceil_id = (int)std::floor((m->cfg.virtual_ceil_height - g.org[2]) * g.res_inv);
    if (ceil_id >= 0 && ceil_id < g.nz) {
      int n = (b.hi[0] - b.lo[0] + 1) * (b.hi[1] - b.lo[1] + 1);
      // (the ceiling rewrites occupancy-plane words: a search in flight may still read them -- ADVICE r4)
      HIPCHK(map_wait_plane_readers(m));
      k_virtual_ceil<<<blocks_for(n, 256), 256, 0, m->stream>>>(
          g, b, ceil_id, m->info.clamp_max_log, m->occ, m->occ_bits.p, m->unk_bits.p,
          m->info.min_occupancy_log, m->info.clamp_min_log - 1e-3);
      HIPCHK(hipEventRecord(m->ev_planes, m->stream));
      ++m->planes_ver;
    }
  }
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

extern "C" int fuelmi_map_last_inflate_kernel(const fuelmi_map* m) { return m ? m->last_inflate_kernel : FUELMI_EINVAL; }

extern "C" int fuelmi_map_update_esdf(fuelmi_map* m) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  // Alone on the map's reader / writer lock from the look at the query slots to the LAST launch of the update: a query
  // that recorded its "behind the map's stream" event between two of the update's kernels would wait for the first ones
  // only (signed maps: it would read the positive field while the negative pass merges into it)
  std::unique_lock<std::shared_timed_mutex> wr(m->rw_mu);
  {
    const int rcq = map_wait_query_readers(m);  // (query kernels in flight still read the field this update rewrites)
    if (rcq) return rcq;
  }
  return esdf_update(m);
}
// Every entry point that rewrites the distance field (or the inflated plane the slot kernels' callers pair it with) is a
// WRITER in the sense of the contract in fuelmi.h: alone on rw_mu from the look at the query slots to its last launch,
// behind the query kernels already launched (the resets below wrote dist without either).
#define MAP_WRITER_PROLOGUE(m)                                  \
  std::unique_lock<std::shared_timed_mutex> wr__((m)->rw_mu);   \
  {                                                             \
    const int rcq__ = map_wait_query_readers(m);                \
    if (rcq__) return rcq__;                                    \
  }

extern "C" int fuelmi_map_reset_buffer(fuelmi_map* m, const double min_pos[3], const double max_pos[3]) {
  if (m) ++m->occ_epoch;
  ARGCHK(m && min_pos && max_pos);
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  Box3 b;
  pos_to_index(m, min_pos, b.lo);
  pos_to_index(m, max_pos, b.hi);
  bound_index(m, b.lo);
  bound_index(m, b.hi);
  for (int i = 0; i < 3; ++i)
    if (b.lo[i] > b.hi[i]) return FUELMI_OK;
  int w_lo = (int)(adr_of(g, b.lo) >> 6), w_hi = (int)(adr_of(g, b.hi) >> 6);
  MAP_WRITER_PROLOGUE(m);
  k_reset_bits<<<blocks_for(w_hi - w_lo + 1, 256), 256, 0, m->stream>>>(g, b, m->infl_bits.p, w_lo, w_hi);
  long n = (long)(b.hi[0] - b.lo[0] + 1) * (b.hi[1] - b.lo[1] + 1) * (b.hi[2] - b.lo[2] + 1);
  k_reset_dist<<<blocks_for(n, 256, 65536), 256, 0, m->stream>>>(g, b, m->dist, (float)m->cfg.default_dist);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

extern "C" int fuelmi_map_reset_buffer_all(fuelmi_map* m) {
  ARGCHK(m);
  int rc = fuelmi_map_reset_buffer(m, m->info.min_boundary, m->info.max_boundary);
  if (rc) return rc;
  for (int i = 0; i < 3; ++i) m->local_bound.lo[i] = 0;
  m->local_bound.hi[0] = m->g.nx - 1;
  m->local_bound.hi[1] = m->g.ny - 1;
  m->local_bound.hi[2] = m->g.nz - 1;
  return FUELMI_OK;
}

extern "C" int fuelmi_map_set_occupied(fuelmi_map* m, const double* pos, int n, int occ) {
  ARGCHK(m && (pos || n == 0) && n >= 0);
  if (occ != 0 && occ != 1) {
    fuelmi_set_error("setOccupied: only occ in {0,1} is representable (inflate buffer is a bit-plane)");
    return FUELMI_ELIMIT;
  }
  if (n == 0) return FUELMI_OK;
  HIPCHK(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lk(m->qmu);
  size_t bytes = (size_t)n * 3 * sizeof(double);
  int rc = map_ensure_stage(m, bytes, 0);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(m->d_stage, pos, bytes, hipMemcpyHostToDevice, m->stream));
  k_set_occupied<<<blocks_for(n, 256), 256, 0, m->stream>>>(m->g, (const double*)m->d_stage, n, occ,
                                                            m->infl_bits.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(m->stream));
  return FUELMI_OK;
}

extern "C" int fuelmi_map_get_local_bound(const fuelmi_map* m, int bmin[3], int bmax[3]) {
  ARGCHK(m && bmin && bmax);
  for (int i = 0; i < 3; ++i) bmin[i] = m->local_bound.lo[i], bmax[i] = m->local_bound.hi[i];
  return FUELMI_OK;
}
extern "C" int fuelmi_map_set_local_bound(fuelmi_map* m, const int bmin[3], const int bmax[3]) {
  ARGCHK(m && bmin && bmax);
  const int nv[3] = {m->g.nx, m->g.ny, m->g.nz};
  for (int i = 0; i < 3; ++i) ARGCHK(bmin[i] >= 0 && bmax[i] < nv[i] && bmin[i] <= bmax[i]);
  for (int i = 0; i < 3; ++i) m->local_bound.lo[i] = bmin[i], m->local_bound.hi[i] = bmax[i];
  return FUELMI_OK;
}
extern "C" int fuelmi_map_get_updated_box(fuelmi_map* m, double bmin[3], double bmax[3], int reset) {
  ARGCHK(m && bmin && bmax);
  for (int i = 0; i < 3; ++i) bmin[i] = m->upd_min[i], bmax[i] = m->upd_max[i];
  if (reset) m->reset_updated_box = true;
  return FUELMI_OK;
}
extern "C" int fuelmi_map_set_updated_box(fuelmi_map* m, const double bmin[3], const double bmax[3]) {
  ARGCHK(m && bmin && bmax);
  for (int i = 0; i < 3; ++i) m->upd_min[i] = bmin[i], m->upd_max[i] = bmax[i];
  m->reset_updated_box = false;
  return FUELMI_OK;
}

extern "C" int fuelmi_map_register_mirrors(fuelmi_map* m, double* occupancy, char* inflate, double* distance) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lk(m->qmu);
  HIPCHK(hipStreamSynchronize(m->stream));
  void* host[3] = {occupancy, inflate, distance};
  const size_t bytes[3] = {(size_t)m->g.N * sizeof(double), (size_t)m->g.N, (size_t)m->g.N * sizeof(double)};
  for (int k = 0; k < 3; ++k) {
    fuelmi_map::Mirror& r = m->mirror[k];
    if (r.host && r.host != host[k]) {
      (void)hipHostUnregister(r.host);
      r.host = r.dev = nullptr;
    }
    if (!host[k] || r.host == host[k]) continue;
    // pin the caller's buffer where it lies and map it: the refresh kernel then stores straight into it
    HIPCHK(hipHostRegister(host[k], bytes[k], hipHostRegisterMapped));
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host[k], 0) != hipSuccess) {
      (void)hipHostUnregister(host[k]);
      fuelmi_set_error("hipHostGetDevicePointer failed for a mirror buffer");
      return FUELMI_EHIP;
    }
    r.host = host[k];
    r.dev = d;
  }
  return FUELMI_OK;
}
extern "C" int fuelmi_map_unregister_mirrors(fuelmi_map* m) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lk(m->qmu);
  HIPCHK(hipStreamSynchronize(m->stream));
  for (auto& r : m->mirror) {
    if (r.host) (void)hipHostUnregister(r.host);
    r.host = r.dev = nullptr;
  }
  return FUELMI_OK;
}

extern "C" int fuelmi_map_sync_host(fuelmi_map* m, const int bmin[3], const int bmax[3], double* occupancy,
                                    char* inflate, double* distance) {
  ARGCHK(m);
  if (!occupancy && !inflate && !distance) return FUELMI_OK;
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  Box3 b;
  const int nv[3] = {g.nx, g.ny, g.nz};
  for (int k = 0; k < 3; ++k) b.lo[k] = 0, b.hi[k] = nv[k] - 1;
  if (bmin && bmax)
    for (int k = 0; k < 3; ++k) {
      ARGCHK(bmin[k] >= 0 && bmax[k] < nv[k] && bmin[k] <= bmax[k]);
      b.lo[k] = bmin[k], b.hi[k] = bmax[k];
    }
  const int xlen = b.hi[0] - b.lo[0] + 1, ylen = b.hi[1] - b.lo[1] + 1, zlen = b.hi[2] - b.lo[2] + 1;
  const long total = (long)xlen * ylen * zlen;
  std::lock_guard<std::mutex> lk(m->qmu);
  // (1) registered mirrors: one kernel stores the box voxels straight into the caller's pinned buffers
  void* want[3] = {occupancy, inflate, distance};
  void* dir[3] = {nullptr, nullptr, nullptr};
  bool any_direct = false, any_staged = false;
  for (int k = 0; k < 3; ++k) {
    if (!want[k]) continue;
    if (m->mirror[k].host == want[k])
      dir[k] = m->mirror[k].dev, any_direct = true;
    else
      any_staged = true;
  }
  if (any_direct)
    k_sync_box<true><<<blocks_for(total, 256, 8192), 256, 0, m->stream>>>(
        g, b, m->occ, m->infl_bits.p, m->dist, (double*)dir[0], (char*)dir[1], (double*)dir[2]);
  // (2) other buffers (pageable memory): compact box arrays -> pinned staging in ONE copy -> rows scattered
  // by the host; the PCIe traffic is the box, not the x-slabs around it
  unsigned char* h = nullptr;
  size_t off[3] = {0, 0, 0};
  if (any_staged) {
    const size_t esz[3] = {sizeof(double), 1, sizeof(double)};
    size_t bytes = 0;
    for (int k = 0; k < 3; ++k)
      if (want[k] && !dir[k]) off[k] = bytes, bytes += (((size_t)total * esz[k]) + 255) & ~(size_t)255;
    int rc = map_ensure_stage(m, bytes, bytes);
    if (rc) return rc;
    unsigned char* d = reinterpret_cast<unsigned char*>(m->d_stage);
    k_sync_box<false><<<blocks_for(total, 256, 8192), 256, 0, m->stream>>>(
        g, b, m->occ, m->infl_bits.p, m->dist, (want[0] && !dir[0]) ? (double*)(d + off[0]) : nullptr,
        (want[1] && !dir[1]) ? (char*)(d + off[1]) : nullptr, (want[2] && !dir[2]) ? (double*)(d + off[2]) : nullptr);
    HIPCHK(hipMemcpyAsync(m->h_stage, m->d_stage, bytes, hipMemcpyDeviceToHost, m->stream));
    h = reinterpret_cast<unsigned char*>(m->h_stage);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(m->stream));  // the only wait
  if (any_staged) {
    const bool whole_rows = zlen == g.nz;  // rows of a full-height box are contiguous per x (or altogether)
    for (int k = 0; k < 3; ++k) {
      if (!want[k] || dir[k]) continue;
      const size_t e = k == 1 ? 1 : sizeof(double);
      unsigned char* dst = reinterpret_cast<unsigned char*>(want[k]);
      const unsigned char* src = h + off[k];
      if (whole_rows && ylen == g.ny) {
        memcpy(dst + (size_t)b.lo[0] * g.nyz * e, src, (size_t)total * e);
        continue;
      }
      for (int xi = 0; xi < xlen; ++xi)
        for (int yi = 0; yi < ylen; ++yi) {
          const size_t a = (size_t)(b.lo[0] + xi) * g.nyz + (size_t)(b.lo[1] + yi) * g.nz + b.lo[2];
          memcpy(dst + a * e, src + ((size_t)xi * ylen + yi) * zlen * e, (size_t)zlen * e);
        }
    }
  }
  return FUELMI_OK;
}

extern "C" int fuelmi_map_dist_grad(fuelmi_map* m, const double* pos, int n, double* dist, double* grad) {
  ARGCHK(m && n >= 0 && (n == 0 || (pos && dist && grad)));
  if (n == 0) return FUELMI_OK;
  HIPCHK(hipSetDevice(m->device));
  // re-entrant: a query slot -- own side stream, positions and results in its pinned block
  const size_t in_b = (size_t)n * 3 * sizeof(double), out_b = (size_t)n * 4 * sizeof(double);
  QuerySlotGuard q;
  const int rc = q.acquire(m, in_b + out_b);
  if (rc) return rc;
  double* h_pos = reinterpret_cast<double*>(q.s->pin);
  double* h_d = h_pos + (size_t)n * 3;
  double* h_g = h_d + n;
  memcpy(h_pos, pos, in_b);
  k_dist_grad<<<blocks_for(n, 128), 128, 0, q.s->st>>>(m->g, m->dist, h_pos, n, h_d, h_g);
  HIPCHK(hipGetLastError());
  HIPCHK(q.finish());
  memcpy(dist, h_d, (size_t)n * sizeof(double));
  memcpy(grad, h_g, (size_t)n * 3 * sizeof(double));
  return FUELMI_OK;
}

extern "C" int fuelmi_map_coarse_dist(fuelmi_map* m, const double* pos, int n, double* dist) {
  ARGCHK(m && n >= 0 && (n == 0 || (pos && dist)));
  if (n == 0) return FUELMI_OK;
  HIPCHK(hipSetDevice(m->device));
  const size_t in_b = (size_t)n * 3 * sizeof(double);
  QuerySlotGuard q;
  const int rc = q.acquire(m, in_b + (size_t)n * sizeof(double));
  if (rc) return rc;
  double* h_pos = reinterpret_cast<double*>(q.s->pin);
  double* h_d = h_pos + (size_t)n * 3;
  memcpy(h_pos, pos, in_b);
  k_coarse_dist<<<blocks_for(n, 128), 128, 0, q.s->st>>>(m->g, m->dist, h_pos, n, h_d);
  HIPCHK(hipGetLastError());
  HIPCHK(q.finish());
  memcpy(dist, h_d, (size_t)n * sizeof(double));
  return FUELMI_OK;
}

extern "C" int fuelmi_map_query_state(fuelmi_map* m, const int* idx, int n, int* occupancy, int* inflate) {
  ARGCHK(m && n >= 0 && (n == 0 || (idx && occupancy && inflate)));
  if (n == 0) return FUELMI_OK;
  HIPCHK(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lk(m->qmu);
  size_t in_b = (size_t)n * 3 * sizeof(int);
  int rc = map_ensure_stage(m, in_b + (size_t)n * 2 * sizeof(int), 0);
  if (rc) return rc;
  int* d_idx = (int*)m->d_stage;
  int* d_o = d_idx + (size_t)n * 3;
  int* d_i = d_o + n;
  HIPCHK(hipMemcpyAsync(d_idx, idx, in_b, hipMemcpyHostToDevice, m->stream));
  k_query_state<<<blocks_for(n, 128), 128, 0, m->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, m->infl_bits.p,
                                                           d_idx, n, d_o, d_i);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(occupancy, d_o, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipMemcpyAsync(inflate, d_i, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  return FUELMI_OK;
}

extern "C" int fuelmi_map_synchronize(fuelmi_map* m) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(stream_wait(m->stream));
  return FUELMI_OK;
}

// ---------------------------------------------------------------------------------------------
// measurement hooks
// ---------------------------------------------------------------------------------------------
extern "C" int fuelmi_timer_begin(fuelmi_map* m) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipEventRecord(m->t0, m->stream));
  return FUELMI_OK;
}
extern "C" int fuelmi_timer_end(fuelmi_map* m, float* ms) {
  ARGCHK(m && ms);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipEventRecord(m->t1, m->stream));
  HIPCHK(hipEventSynchronize(m->t1));
  HIPCHK(hipEventElapsedTime(ms, m->t0, m->t1));
  return FUELMI_OK;
}
extern "C" int fuelmi_profile_enable(fuelmi_map* m, unsigned mask) {
  ARGCHK(m);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  m->profile_mask = mask;
  for (auto& s : m->prof) s.used = 0;
  if (mask) HIPCHK(hipEventRecord(m->t_prof0, m->stream));  // the origin of fuelmi_profile_get_timeline (its own event: the
                                                          // timer's t0 moves with every fuelmi_timer_begin -- ADVICE r5)
  return FUELMI_OK;
}
// Begin and end of every bracket of a stage in milliseconds after the fuelmi_profile_enable call that armed it: the
// device's own timeline of a few cycles without a tracer attached to the host (a kernel trace slows the issuing thread
// and stretches exactly the gaps one wants to see).  All events share the device's clock, whatever stream they were
// recorded on.
extern "C" int fuelmi_profile_get_timeline(fuelmi_map* m, int stage, double* begin_ms, double* end_ms, int cap, int* n) {
  ARGCHK(m && stage >= 0 && stage < FUELMI_K_COUNT && begin_ms && end_ms && cap >= 0 && n);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipEventSynchronize(m->t_prof0));
  ProfileSlot& s = m->prof[stage];
  int k = 0;
  for (size_t i = 0; i + 1 < s.used && k < cap; i += 2, ++k) {
    float a = 0.f, b = 0.f;
    HIPCHK(hipEventSynchronize(s.ev[i + 1]));
    HIPCHK(hipEventElapsedTime(&a, m->t_prof0, s.ev[i]));
    HIPCHK(hipEventElapsedTime(&b, m->t_prof0, s.ev[i + 1]));
    begin_ms[k] = a, end_ms[k] = b;
  }
  *n = k;
  return FUELMI_OK;
}
extern "C" int fuelmi_profile_get(fuelmi_map* m, int stage, int* launches, double* total_ms) {
  ARGCHK(m && stage >= 0 && stage < FUELMI_K_COUNT && launches && total_ms);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  ProfileSlot& s = m->prof[stage];
  double tot = 0.0;
  for (size_t i = 0; i + 1 < s.used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventSynchronize(s.ev[i + 1]));  // stages of the frontier finder are recorded on ITS stream
    HIPCHK(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
    tot += ms;
  }
  *launches = (int)(s.used / 2);
  *total_ms = tot;
  return FUELMI_OK;
}
extern "C" int fuelmi_profile_get_samples(fuelmi_map* m, int stage, double* ms_out, int cap, int* n) {
  ARGCHK(m && stage >= 0 && stage < FUELMI_K_COUNT && ms_out && cap >= 0 && n);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  ProfileSlot& s = m->prof[stage];
  int k = 0;
  for (size_t i = 0; i + 1 < s.used && k < cap; i += 2, ++k) {
    float ms = 0.f;
    HIPCHK(hipEventSynchronize(s.ev[i + 1]));
    HIPCHK(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
    ms_out[k] = ms;
  }
  *n = k;
  return FUELMI_OK;
}
