// frontier.hip -- frontier detection + clustering (FrontierFinder::searchFrontiers /
// expandFrontier, active_perception/src/frontier_finder.cpp:54-164; predicates :862-877;
// haveOverlap / isFrontierChanged / computeFrontierInfo :353-390).
//
// The reference scans the search box in x,y,z order and region-grows (26-neighbourhood BFS) from
// every unflagged frontier cell.  The result of that SEQUENTIAL process is reproduced exactly by
// an order-independent formulation:
//   F1(c)   = FREE(c) and one of the 6 face neighbours is UNKNOWN            (:862-877)
//   Q0      = F1 & flag==0 & isInBox(idx) & centre.z >= min_z   cells the BFS may ADD (:146-151)
//   seeds   = F1 & flag==0 & inside the scanned index box       cells the scan may START from
//   NQ seed = seed & !Q0 (below min_z / on the box_max face): starts a cluster, is never added.
// BFS growth only walks Q0 cells, so clusters are unions of 26-connected components of Q0.
// A component C is claimed by the first (lowest address = scan order) of: its own cells inside
// the scan box, or the NQ seeds adjacent to it.  cluster(C) = that claimer; every NQ seed also
// forms a cluster with the components it claims.  Flags are set for every claimed cell and every
// NQ seed, kept or not (the reference's sticky flags of rejected small clusters, :136,154-163).
//
// Device pipeline: bit-plane predicate per 64-voxel word (funnel-shifted neighbour planes, wave
// prefix sums) -> ordered compaction -> lock-free union-find over the compact cells (neighbour
// lookup = bit test + popcount rank) -> atomicMin claims -> sizes -> flags.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <unordered_map>
#include <vector>

#include "fuelmi_internal.h"
#include <functional>
#include <condition_variable>
#include <thread>

#include "frontier_internal.h"

// FUELMI_DEBUG_SYNC=1: synchronise and name every frontier kernel (locates device faults)
// a finder whose map was destroyed first (garbage collectors and destructor orders do that) only accepts _destroy
#define FRONTIER_HAS_MAP(f)                                                                        \
  do {                                                                                             \
    if (!(f)->map) {                                                                               \
      fuelmi_set_error("the map of this frontier finder has been destroyed: only fuelmi_frontier_destroy is legal"); \
      return FUELMI_EINVAL;                                                                        \
    }                                                                                              \
  } while (0)
#define FDBG(name)                                                                     \
  do {                                                                                 \
    static const bool on__ = getenv("FUELMI_DEBUG_SYNC") != nullptr;                   \
    if (on__) {                                                                        \
      hipError_t e__ = hipStreamSynchronize(f->stream);                                \
      std::fprintf(stderr, "[fuelmi] %s: %s\n", name, hipGetErrorString(e__));         \
    }                                                                                  \
  } while (0)

// ---- per-word masks ---------------------------------------------------------------------------
__device__ __forceinline__ void word_masks(const Geo& g, int w, const Box3& qb, const Box3& sb, u64& z0,
                                           u64& zl, u64& y0, u64& yl, u64& mq, u64& ms) {
  z0 = zl = y0 = yl = mq = ms = 0ull;
  long a0 = 64L * w;
  if (a0 >= g.N) return;
  int line = (int)(a0 / g.nz);
  int z = (int)(a0 - (long)line * g.nz);
  int x = line / g.ny;
  int y = line - x * g.ny;
  int bpos = 0;
  while (bpos < 64 && x < g.nx) {
    int len = min(g.nz - z, 64 - bpos);
    if (z == 0) z0 |= 1ull << bpos;
    if (z + len == g.nz) zl |= 1ull << (bpos + len - 1);
    u64 seg = bit_range(bpos, len);
    if (y == 0) y0 |= seg;
    if (y == g.ny - 1) yl |= seg;
    if (x >= qb.lo[0] && x <= qb.hi[0] && y >= qb.lo[1] && y <= qb.hi[1]) {
      int zlo = max(z, qb.lo[2]), zhi = min(z + len - 1, qb.hi[2]);
      if (zlo <= zhi) mq |= bit_range(bpos + (zlo - z), zhi - zlo + 1);
    }
    if (x >= sb.lo[0] && x <= sb.hi[0] && y >= sb.lo[1] && y <= sb.hi[1]) {
      int zlo = max(z, sb.lo[2]), zhi = min(z + len - 1, sb.hi[2]);
      if (zlo <= zhi) ms |= bit_range(bpos + (zlo - z), zhi - zlo + 1);
    }
    bpos += len;
    z = 0;
    if (++y == g.ny) {
      y = 0;
      ++x;
    }
  }
}

// F1 for the 64 voxels of word w (knownfree && isNeighborUnknown); out-of-map neighbours are
// "-1", i.e. not UNKNOWN (sdf_map.h:196-198)
__device__ __forceinline__ u64 f1_word(const Geo& g, const u64* __restrict__ occ, const u64* __restrict__ unk,
                                       int w, u64 z0, u64 zl, u64 y0, u64 yl) {
  long a0 = 64L * w;
  u64 valid = (a0 + 64 <= g.N) ? ~0ull : bit_range(0, (int)max(0L, g.N - a0));
  u64 free_ = ~occ[w] & ~unk[w] & valid;
  if (free_ == 0ull) return 0ull;
  u64 nb = (plane_window(unk, a0 + 1) & ~zl) | (plane_window(unk, a0 - 1) & ~z0) |
           (plane_window(unk, a0 + g.nz) & ~yl) | (plane_window(unk, a0 - g.nz) & ~y0) |
           plane_window(unk, a0 + g.nyz) | plane_window(unk, a0 - g.nyz);
  return free_ & nb;
}

__device__ __forceinline__ bool f1_cell(const Geo& g, const u64* __restrict__ occ, const u64* __restrict__ unk,
                                        long a) {
  auto bit = [&](const u64* p, long q) { return (p[q >> 6] >> (q & 63)) & 1ull; };
  if (bit(occ, a) || bit(unk, a)) return false;
  int x = (int)(a / g.nyz);
  int r = (int)(a - (long)x * g.nyz);
  int y = r / g.nz, z = r - y * g.nz;
  if (x > 0 && bit(unk, a - g.nyz)) return true;
  if (x < g.nx - 1 && bit(unk, a + g.nyz)) return true;
  if (y > 0 && bit(unk, a - g.nz)) return true;
  if (y < g.ny - 1 && bit(unk, a + g.nz)) return true;
  if (z > 0 && bit(unk, a - 1)) return true;
  if (z < g.nz - 1 && bit(unk, a + 1)) return true;
  return false;
}

// committed clusters' cells live in one device pool; candidate k of a changed-cluster test owns the
// global indices [start[k], start[k+1]).  The (pool offset, start) table is read straight from the pinned host
// copy and staged in LDS (no H2D copy node in front of the search), verdicts go straight back to pinned memory.
struct RmCand {
  u64 off;    // pool offset of the cluster's cells
  u32 start;  // first flat index
  u32 pad;
};
#define RM_LDS 1024  // candidates staged per block
__device__ __forceinline__ int rm_cluster_of(const u32* s_start, int ncand, u32 i) {
  int lo = 0, hi = ncand - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s_start[mid] <= i)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}
// MODE 0: "did a cell stop being a frontier cell?" -> d_mark[k] = mark, h_changed[k] = 1
// MODE 1: clear the flags of the clusters marked by MODE 0
template <int MODE>
__global__ void __launch_bounds__(256)
k_rm_pool(Geo g, const u64* __restrict__ occ, const u64* __restrict__ unk, u64* flag, const u32* __restrict__ pool,
          const RmCand* __restrict__ cand, int ncand, u32 total, int* d_mark, int mark, int* h_changed) {
  __shared__ u32 s_start[RM_LDS];
  __shared__ u64 s_off[RM_LDS];
  const bool staged = ncand <= RM_LDS;  // else `cand` is a device copy and the look-ups go to memory
  if (staged) {
    for (int k = threadIdx.x; k < ncand; k += 256) {
      const RmCand c = cand[k];
      s_start[k] = c.start;
      s_off[k] = c.off;
    }
    __syncthreads();
  }
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int k;
  u32 a;
  if (staged) {
    k = rm_cluster_of(s_start, ncand, i);
    a = pool[s_off[k] + (i - s_start[k])];
  } else {
    int lo = 0, hi = ncand - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (cand[mid].start <= i)
        lo = mid;
      else
        hi = mid - 1;
    }
    k = lo;
    a = pool[cand[k].off + (i - cand[k].start)];
  }
  if (MODE == 0) {
    if (!f1_cell(g, occ, unk, a) && d_mark[k] != mark) {
      d_mark[k] = mark;
      h_changed[k] = 1;
    }
  } else {
    if (d_mark[k] == mark) atomicAnd(&flag[a >> 6], ~(1ull << (a & 63)));
  }
}
// Both modes in ONE launch of many workgroups (round 6; a streaming frame's changed-cluster test walks a 20 k-cell
// surface: too long for one workgroup, and two dependent launches stood at the head of the frame's critical loop):
// the marks leave with agent-scope stores, every workgroup counts itself in behind s_waitcnt vmcnt(0) and spins until
// the count reaches `target` (the host keeps the running total: nobody resets the counter), then clears the flags of
// the marked clusters, reading the marks past its L2.  The grid is capped at RM_BAR_BLOCKS workgroups of 256 lanes --
// far below what the chip holds at once, so every workgroup is resident (or becomes so as other kernels drain) while
// the others spin: no deadlock.  No fence anywhere.
#define RM_BAR_BLOCKS 512
__global__ void __launch_bounds__(256)
k_rm_pool_bar(Geo g, const u64* __restrict__ occ, const u64* __restrict__ unk, u64* flag, const u32* __restrict__ pool,
              const RmCand* __restrict__ cand, int ncand, u32 total, int* d_mark, int mark, int* h_changed, u32* bar, u32 target) {
  __shared__ u32 s_start[RM_LDS];
  __shared__ u64 s_off[RM_LDS];
  for (int k = threadIdx.x; k < ncand; k += 256) {  // (ncand <= RM_LDS: checked by the host)
    const RmCand c = cand[k];
    s_start[k] = c.start;
    s_off[k] = c.off;
  }
  __syncthreads();
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  int k = 0;
  u32 a = 0u;
  bool chg = false;
  if (i < total) {
    k = rm_cluster_of(s_start, ncand, i);
    a = pool[s_off[k] + (i - s_start[k])];
    chg = !f1_cell(g, occ, unk, a);
  }
  {  // one mark per wave and cluster (a wave's cells nearly always belong to one cluster: the verdict crosses PCIe)
    const unsigned long long bm = __ballot(chg);
    if (bm) {
      const int leader = __builtin_ctzll(bm);
      const int kl = __shfl(k, leader, 64);
      if (chg && ((int)(threadIdx.x & 63) == leader || k != kl)) {
        st_agent(reinterpret_cast<u32*>(d_mark) + k, (u32)mark);
        h_changed[k] = 1;
      }
    }
  }
  wait_vm_stores();
  __syncthreads();
  // arrival: ONE returning atomic per workgroup; the last arriver releases everybody through per-workgroup words 64 bytes
  // apart (pollers of one word serialise at the memory-side atomic unit and starve the arrivals: measured, 9 % of a frame)
  __shared__ u32 s_lastw;
  if (threadIdx.x == 0) s_lastw = atomicAdd(bar, 1u) + 1u == target ? 1u : 0u;
  __syncthreads();
  if (s_lastw) {
    for (u32 w = threadIdx.x; w < gridDim.x; w += 256) st_agent(bar + 16u * (w + 1u), (u32)mark);
  } else if (threadIdx.x == 0) {
    const u32* mine = bar + 16u * (blockIdx.x + 1u);
    const unsigned long long t0 = wall_clock64();
    for (u32 spins = 0; ld_agent(mine) != (u32)mark; ++spins) {
      if ((spins & 63u) == 63u && wall_clock64() - t0 > 5000000ull) {  // 50 ms: unreachable with a resident grid -- never
        h_changed[ncand] = -1;                                          // hang the device, and never finish silently:
        break;                                                          // _search_end reports the search as failed
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (i < total && ld_agent(reinterpret_cast<const u32*>(d_mark) + k) == (u32)mark) atomicAnd(&flag[a >> 6], ~(1ull << (a & 63)));
}
// both modes in ONE workgroup for the searches of an exploration (a few thousand pooled cells in the clusters the updated
// box touches: one launch instead of two dependent ones -- the first kernels of a streaming frame's critical path).
// The marks live in LDS; h_changed[k] is written as by MODE 0.
#define RM_ONE_T 1024
#define RM_ONE_CELLS (2 * RM_ONE_T)  // (round 6: was 16 per lane -- a streaming frame's 10-16 k-cell candidates kept ONE workgroup busy for 13-77 us,
                                     // profiles/r06_rm_pool_variants.txt; beyond two cells per lane k_rm_pool_bar takes the test)
__global__ void __launch_bounds__(RM_ONE_T)
k_rm_pool_one(Geo g, const u64* __restrict__ occ, const u64* __restrict__ unk, u64* flag, const u32* __restrict__ pool,
              const RmCand* __restrict__ cand, int ncand, u32 total, int* h_changed) {
  __shared__ u32 s_start[RM_LDS];
  __shared__ u64 s_off[RM_LDS];
  __shared__ u32 s_mark[RM_LDS];
  for (int k = threadIdx.x; k < ncand; k += RM_ONE_T) {
    const RmCand c = cand[k];
    s_start[k] = c.start;
    s_off[k] = c.off;
    s_mark[k] = 0u;
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < total; i += RM_ONE_T) {
    const int k = rm_cluster_of(s_start, ncand, i);
    const u32 a = pool[s_off[k] + (i - s_start[k])];
    if (!f1_cell(g, occ, unk, a) && s_mark[k] == 0u) {
      s_mark[k] = 1u;
      h_changed[k] = 1;
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < total; i += RM_ONE_T) {
    const int k = rm_cluster_of(s_start, ncand, i);
    if (s_mark[k]) {
      const u32 a = pool[s_off[k] + (i - s_start[k])];
      atomicAnd(&flag[a >> 6], ~(1ull << (a & 63)));
    }
  }
}
__global__ void k_pool_put(u32* __restrict__ pool, const u32* __restrict__ cells, const PoolPut* __restrict__ table) {
  const PoolPut e = table[blockIdx.x];  // one workgroup per cluster (the table sits in pinned host memory)
  u32* dst = pool + e.dst;
  const u32* src = cells + e.src;
  for (u32 i = threadIdx.x; i < e.n; i += blockDim.x) dst[i] = src[i];
  if (e.seed >= 0 && threadIdx.x == 0) dst[e.n] = (u32)e.seed;  // order is irrelevant on the device
}
__device__ void scan_sums_tail(const FArgs& F) {
  // one block of 1024 threads: contiguous slices of the block sums per thread, then a Hillis-Steele scan
  // of the 1024 partials in LDS (a 256-thread version with a serial middle cost 18 us on an 800^2 x 200 map)
  __shared__ u64 part[1024];
  const int nblocks = F.var->nblocks;
  const int T = blockDim.x;
  const int per = (nblocks + T - 1) / T;
  const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
  u64 s = 0;
  for (int b = b0; b < b1; ++b) s += F.blocksum[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {
    const u64 v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  if (threadIdx.x == T - 1) {
    const u64 run = part[T - 1];
    u32 nq = (u32)run, ns = (u32)(run >> 32);
    u32 ovf = 0;
    if (nq > F.cap_q) {
      nq = F.cap_q;
      ovf = 1;
    }
    if (ns > F.cap_s) {
      ns = F.cap_s;
      ovf = 1;
    }
    F.counts[0] = nq;
    F.counts[1] = ns;
    F.counts[2] = ovf;
    F.counts[3] = 0;
    F.counts[5] = 0;
  }
  u64 run = part[threadIdx.x] - s;  // exclusive prefix of this thread's slice
  for (int b = b0; b < b1; ++b) {
    u64 v = F.blocksum[b];
    F.blockscan[b] = run;
    run += v;
  }
}

// predicate planes + in-block packed prefix of their popcounts
__global__ void __launch_bounds__(256) k_pred(Geo g, FArgs F) {
  __shared__ u64 wsum[4];
  const FVar& V = *F.var;
  if ((int)blockIdx.x >= V.nblocks) return;
  const int rel = blockIdx.x * 256 + threadIdx.x;
  const int w = V.w0 + rel;
  u64 q = 0ull, s = 0ull;
  if (w < g.W) {
    u64 z0, zl, y0, yl, mq, ms;
    word_masks(g, w, V.qreg, V.sbox, z0, zl, y0, yl, mq, ms);
    if ((mq | ms) != 0ull) {
      u64 f1 = f1_word(g, F.occ, F.unk, w, z0, zl, y0, yl) & ~F.flag[w];
      q = f1 & mq;
      s = f1 & ms & ~mq;
    }
    F.qb[w] = q;
    F.sb[w] = s;
  }
  u64 packed = (u64)__popcll(q) | ((u64)__popcll(s) << 32);
  u64 v = packed;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    u64 t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  u64 woff = 0;
  for (int k = 0; k < wave; ++k) woff += wsum[k];
  u64 excl = v - packed + woff;
  F.pref[rel] = excl;
  if (threadIdx.x == 255) F.blocksum[blockIdx.x] = excl + packed;
}
__global__ void __launch_bounds__(1024) k_scan_sums(FArgs F) { scan_sums_tail(F); }

// ordered compaction of Q0 cells and NQ seeds
__global__ void __launch_bounds__(256) k_compact(Geo g, FArgs F) {
  if ((int)blockIdx.x >= F.var->nblocks) return;
  const int rel = blockIdx.x * 256 + threadIdx.x;
  const int w = F.var->w0 + rel;
  if (w >= g.W) return;
  u64 q = F.qb[w], s = F.sb[w];
  if ((q | s) == 0ull) return;
  u64 pk = F.blockscan[rel >> 8] + F.pref[rel];
  u32 iq = (u32)pk, is = (u32)(pk >> 32);
  const u32 nq = F.counts[0];
  while (q) {
    int b = __builtin_ctzll(q);
    q &= q - 1;
    if (iq < F.cap_q) {
      F.cell_adr[iq] = (u32)(64L * w + b);
      F.parent[iq] = iq;
      F.claim[iq] = NOCLAIM;
      F.csize[iq] = 0;
    }
    ++iq;
  }
  while (s) {
    int b = __builtin_ctzll(s);
    s &= s - 1;
    if (is < F.cap_s) {
      F.seed_adr[is] = (u32)(64L * w + b);
      F.csize[nq + is] = 1;  // every NQ seed starts its own cluster
    }
    ++is;
  }
}

// find with path halving.  The shortcut is written with a (non-returning) device-scope atomicMin,
// NOT a plain store: plain stores stay dirty in this XCD's write-back L2 and their later line
// write-back would clobber links other XCDs made on neighbouring entries with memory-side atomics
// (observed as lost unions).  Every write to parent[] inside k_union is therefore an atomic; plain
// reads may be stale, but a stale parent is still an ancestor (parents only decrease).
__device__ __forceinline__ u32 uf_find(u32* parent, u32 i) {
  u32 p = parent[i];
  while (p != i) {
    u32 gp = parent[p];
    if (gp != p) (void)__hip_atomic_fetch_min(&parent[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i = p;
    p = gp;
  }
  return i;
}
__device__ __forceinline__ void uf_union(u32* parent, u32 a, u32 b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      u32 t = a;
      a = b;
      b = t;
    }
    u32 old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;
  }
}

// ---- connected components, 26-connectivity -------------------------------------------------------
// Two levels.  (1) k_ccl_local: a workgroup owns a spatial tile of TX x TY z-lines, labels its Q0
// voxels in LDS (union-find with LDS atomics: ~40 ns per dependent step instead of ~1 us through
// memory-side atomics) and writes parent[cell] = compact index of the tile-local root, so every
// local component leaves the kernel flat.  (2) k_union: only neighbour relations that cross a tile
// face go through the global lock-free union-find.  (3) k_flatten.
#define LNONE 0xFFFFFFFFu
// find with path halving for the long-lived structures of the fast chain (a few thousand tile roots hanging off
// one giant component): the shortcut is an atomicMin, so a concurrent link is never overwritten by a larger value
__device__ __forceinline__ u32 lds_find_h(u32* lab, u32 i) {
  u32 p = reinterpret_cast<volatile u32*>(lab)[i];
  while (p != i) {
    const u32 gp = reinterpret_cast<volatile u32*>(lab)[p];
    if (gp != p) atomicMin(&lab[i], gp);
    i = p;
    p = gp;
  }
  return i;
}
__device__ __forceinline__ void lds_union_h(u32* lab, u32 a, u32 b) {
  while (true) {
    a = lds_find_h(lab, a);
    b = lds_find_h(lab, b);
    if (a == b) return;
    if (a < b) {
      u32 t = a;
      a = b;
      b = t;
    }
    u32 old = atomicMin(&lab[a], b);
    if (old == a) return;
    a = old;
  }
}
__device__ __forceinline__ u32 lds_find(volatile u32* lab, u32 i) {
  u32 p = lab[i];
  while (p != i) {
    i = p;
    p = lab[i];
  }
  return i;
}
__device__ __forceinline__ void lds_union(u32* lab, u32 a, u32 b) {
  while (true) {
    a = lds_find(lab, a);
    b = lds_find(lab, b);
    if (a == b) return;
    if (a < b) {
      u32 t = a;
      a = b;
      b = t;
    }
    u32 old = atomicMin(&lab[a], b);
    if (old == a) return;
    a = old;
  }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_ccl_local(Geo g, FArgs F, int TX, int TY) {
  if ((int)blockIdx.x >= F.var->ntiles) return;
  const Box3& QR = F.var->qreg;
  const int nty = F.var->nty;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32* lab = reinterpret_cast<u32*>(smem_raw);  // [TX*TY lines][nz]
  const int tx = blockIdx.x / nty, ty = blockIdx.x - tx * nty;
  const int x0 = QR.lo[0] + tx * TX, y0 = QR.lo[1] + ty * TY;
  const int nxl = min(TX, QR.hi[0] - x0 + 1), nyl = min(TY, QR.hi[1] - y0 + 1);
  const int nz = g.nz, nseg = (nz + 31) >> 5;
  const int items = TX * TY * nseg;
  u32* segb = lab + TX * TY * nz;  // [TX*TY][nseg] Q0 bits of each 32-voxel segment
  u32* segrank = segb + items;     // compact index of the first Q0 voxel at/after the segment start
  u32* rowr = segrank + items;     // [2*TX] compact index range of each x-row of the tile
  if ((int)threadIdx.x < 2 * TX) {
    const int lx = threadIdx.x >> 1, hi = threadIdx.x & 1;
    u32 rk = 0u;
    if (lx < nxl) rk = min(rank_q(F, (long)(x0 + lx) * g.nyz + (long)(y0 + (hi ? nyl : 0)) * nz), F.cap_q);
    rowr[threadIdx.x] = rk;
  }
  // Everything below walks SET BITS only (frontier cells are ~1 % of the voxels); the only global
  // traffic is this prologue and the parent[] stores at the end.
  for (int it = threadIdx.x; it < items; it += NT) {
    const int line = it / nseg, c = it - line * nseg, lx = line / TY, ly = line - lx * TY;
    const int zn = min(32, nz - 32 * c);
    u32 bits = 0u, rk = 0u;
    if (lx < nxl && ly < nyl) {
      const long lb = (long)(x0 + lx) * g.nyz + (long)(y0 + ly) * nz + 32 * c;
      bits = (u32)plane_window(F.qb, lb);
      if (zn < 32) bits &= (1u << zn) - 1u;
      if (bits) rk = rank_q(F, lb);
    }
    segb[it] = bits;
    segrank[it] = rk;
    const u32 l0 = (u32)(line * nz + 32 * c);
    const u32 all = bits;
    while (bits) {  // 1: label = start of the cell's z-run inside the segment (runs are pre-joined)
      const int z = __builtin_ctz(bits);
      bits &= bits - 1;
      const u32 holes = ~all & ((1u << z) - 1u);
      lab[l0 + z] = l0 + (holes ? 32 - __builtin_clz(holes) : 0);
    }
  }
  __syncthreads();
  // 2 + 3 walk the tile's cells through their COMPACT indices so every lane gets one cell at a time
  // (a vertical frontier wall puts 20-30 cells into one 32-voxel segment; a per-segment loop would
  // leave one lane with all the dependent LDS work).  For a fixed x the TY lines of the tile are
  // contiguous in address, hence contiguous in compact index: TX ranges per tile.
  u32 total = 0u;
  for (int lx = 0; lx < nxl; ++lx) total += rowr[2 * lx + 1] - rowr[2 * lx];
  for (u32 t = threadIdx.x; t < total; t += NT) {
    {
      int lx = 0;
      u32 tt = t;
      for (; lx < nxl - 1; ++lx) {  // which x-row of the tile holds the t-th cell (<= TX steps)
        const u32 n = rowr[2 * lx + 1] - rowr[2 * lx];
        if (tt < n) break;
        tt -= n;
      }
      const u32 i = rowr[2 * lx] + tt;
      const long a_lo = (long)(x0 + lx) * g.nyz + (long)y0 * nz;
      const int rem = (int)(F.cell_adr[i] - (u32)a_lo);  // offset inside this x-row of the tile
      const int ly = rem / nz, z = rem - ly * nz;
      const int line = lx * TY + ly, c = z >> 5, zz = z & 31;
      const u32 v = (u32)(line * nz + z);
      // z-runs are pre-joined inside a segment; join across the segment seam ...
      if (zz == 0 && c > 0 && (segb[line * nseg + c - 1] >> 31)) lds_union(lab, v, v - 1);
      // ... and with the four lower z-lines of the tile: one 3-bit window (dz -1,0,+1) per line, one
      // union per run (only the pattern 101 holds two)
      for (int l = 0; l < 4; ++l) {
        const int nlx = lx + (l < 3 ? -1 : 0), nly = ly + (l < 3 ? l - 1 : -1);
        if (nlx < 0 || nly < 0 || nly >= TY) continue;
        const int nline = nlx * TY + nly;
        const int zlo = z - 1;
        const int s0 = max(zlo, 0) >> 5;
        const unsigned long long w = (unsigned long long)segb[nline * nseg + s0] |
            ((s0 + 1 < nseg) ? ((unsigned long long)segb[nline * nseg + s0 + 1] << 32) : 0ull);
        u32 pat = (zlo >= 0) ? (u32)((w >> (zlo - 32 * s0)) & 7ull) : (u32)((w << 1) & 6ull);
        if (z + 1 >= nz) pat &= 3u;
        if (!pat) continue;
        const u32 ln = (u32)(nline * nz + zlo + __builtin_ctz(pat));
        lds_union(lab, v, ln);
        if (pat == 5u) lds_union(lab, v, ln + 2);
      }
    }
  }
  __syncthreads();
  // 3: parent[cell] = compact index of its tile-local root
  for (u32 t = threadIdx.x; t < total; t += NT) {
    {
      int lx = 0;
      u32 tt = t;
      for (; lx < nxl - 1; ++lx) {
        const u32 n = rowr[2 * lx + 1] - rowr[2 * lx];
        if (tt < n) break;
        tt -= n;
      }
      const u32 i = rowr[2 * lx] + tt;
      const long a_lo = (long)(x0 + lx) * g.nyz + (long)y0 * nz;
      const int rem = (int)(F.cell_adr[i] - (u32)a_lo);
      const int ly = rem / nz, z = rem - ly * nz;
      const u32 v = (u32)((lx * TY + ly) * nz + z);
      const u32 r = lds_find(lab, v);
      u32 pr = i;
      if (r != v) {
        const int rline = (int)(r / (u32)nz), rz = (int)(r - (u32)rline * (u32)nz);
        const int rs = rline * nseg + (rz >> 5);
        pr = segrank[rs] + (u32)__popc(segb[rs] & ((1u << (rz & 31)) - 1u));
      }
      F.parent[i] = pr;
    }
  }
}

// global merge: neighbour relations crossing a tile face (lower-address side only).
// A wall crossing a tile face yields dozens of adjacent cell pairs that all join the same two
// tile-local components; since k_ccl_local left parent[] flat, the pair of tile roots identifies the
// union, and each wave issues one union per DISTINCT root pair (the dependent atomics of the
// lock-free union-find, and the same-address traffic on a huge component's roots, are what costs).
__global__ void __launch_bounds__(256) k_union(Geo g, FArgs F, int TX, int TY) {
  // Per 64 cells: (1) every lane looks up the tile roots behind its cross-face windows (independent
  // loads, all in flight together), (2) the wave removes duplicate root pairs and queues the distinct
  // ones in LDS, (3) the queue is spread over the lanes, one union each.  A union is a chain of
  // dependent ~1 us memory-side operations: what matters is that a wave runs them side by side, not
  // one face direction after the other.
  __shared__ u32 q_a[4][128], q_b[4][128];  // per wave: distinct root pairs of this round
  const u32 nq = F.counts[0];
  const u32 nq_r = (nq + 63u) & ~63u;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nq_r; i += gridDim.x * blockDim.x) {
    const bool live = i < nq;
    // the four lower z-lines (dx,dy) = (-1,-1) (-1,0) (-1,1) (0,-1): one 3-bit window each (dz -1,0,+1);
    // a line inside this cell's tile was already handled in LDS
    const int ldx[4] = {-1, -1, -1, 0}, ldy[4] = {-1, 0, 1, -1};
    u32 pat[4] = {0u, 0u, 0u, 0u};
    long nb0[4] = {0, 0, 0, 0};
    u32 ri = 0u;
    if (live) {
      long a = F.cell_adr[i];
      int x = (int)(a / g.nyz);
      int r = (int)(a - (long)x * g.nyz);
      int y = r / g.nz, z = r - y * g.nz;
      const int lx = (x - F.var->qreg.lo[0]) % TX, ly = (y - F.var->qreg.lo[1]) % TY;
      const bool zlo = z > 0, zhi = z < g.nz - 1;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int xx = x + ldx[l], yy = y + ldy[l];
        const bool cross = (ldx[l] < 0 && lx == 0) || (ldy[l] < 0 && ly == 0) || (ldy[l] > 0 && ly == TY - 1);
        const bool ok = cross && xx >= 0 && yy >= 0 && yy < g.ny;
        nb0[l] = a + (long)ldx[l] * g.nyz + (long)ldy[l] * g.nz - 1;
        u32 p = ok ? (u32)(plane_window(F.qb, nb0[l]) & 7ull) : 0u;
        if (!zlo) p &= ~1u;
        if (!zhi) p &= ~4u;
        pat[l] = p;
      }
      if (pat[0] | pat[1] | pat[2] | pat[3]) ri = F.parent[i];  // tile root (or already an ancestor of it)
    }
    if (!__ballot((pat[0] | pat[1] | pat[2] | pat[3]) != 0u)) continue;
    // z-adjacent neighbours of one line are joined by their own (0,0,-1) unions: one link per run; only
    // the pattern 101 holds two separate runs
    u32 rj[8];
    bool act[8];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      u32 j = F.cap_q;
      if (pat[l]) j = rank_q(F, nb0[l] + __builtin_ctz(pat[l]));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u32 jj = j + (u32)k;
        bool on = ((k == 0) ? (pat[l] != 0u) : (pat[l] == 5u)) && jj < F.cap_q;
        u32 v = 0u;
        if (on) v = F.parent[jj];
        rj[2 * l + k] = v;
        act[2 * l + k] = on;
      }
    }
    u32 nqueue = 0u;  // wave-uniform
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const bool on = act[s8] && rj[s8] != ri;
      u64 todo = __ballot(on);
      bool lead = false;
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const u32 ki = (u32)__shfl((int)ri, leader, 64), kj = (u32)__shfl((int)rj[s8], leader, 64);
        const u64 same = __ballot(on && ri == ki && rj[s8] == kj) & todo;
        if (lane == leader) lead = true;
        todo &= ~same;
      }
      const u64 leads = __ballot(lead);
      if (lead) {
        const u32 slot = nqueue + (u32)__popcll(leads & ((1ull << lane) - 1ull));
        if (slot < 128u) {
          q_a[wv][slot] = ri;
          q_b[wv][slot] = rj[s8];
        } else {
          uf_union(F.parent, ri, rj[s8]);  // queue full (never seen): do it in place
        }
      }
      nqueue += (u32)__popcll(leads);
    }
    nqueue = min(nqueue, 128u);
    __builtin_amdgcn_wave_barrier();
    for (u32 t = lane; t < nqueue; t += 64u) uf_union(F.parent, q_a[wv][t], q_b[wv][t]);
    __builtin_amdgcn_wave_barrier();
  }
}

// root of a cell without writing (k_claim: k_union has finished, parents are final and chains short)
__device__ __forceinline__ u32 uf_root(const u32* parent, u32 i) {
  u32 p = parent[i];
  while (p != i) {
    i = p;
    p = parent[i];
  }
  return i;
}

__device__ __forceinline__ bool in_box(const Geo& g, const Box3& b, long a) {
  int x = (int)(a / g.nyz);
  int r = (int)(a - (long)x * g.nyz);
  int y = r / g.nz, z = r - y * g.nz;
  return x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1] && z >= b.lo[2] && z <= b.hi[2];
}

// one atomicMin per distinct root per wave (a frontier surface is often ONE huge component:
// per-lane atomics on its root serialise)
__device__ __forceinline__ void wave_min_claim(u32* claim, bool active, u32 root, u32 a) {
  u64 todo = __ballot(active);
  const int lane = threadIdx.x & 63;
  while (todo) {
    int leader = __builtin_ctzll(todo);
    u32 k = __shfl(root, leader, 64);
    u64 same = __ballot(active && root == k) & todo;
    u32 v = ((same >> lane) & 1ull) ? a : 0xFFFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) v = min(v, (u32)__shfl_xor((int)v, off, 64));
    if (lane == leader && claim[k] > v) atomicMin(&claim[k], v);
    todo &= ~same;
  }
}

// claims: own cells inside the scan box, then NQ seeds adjacent to a component
__global__ void __launch_bounds__(256) k_claim(Geo g, FArgs F) {
  const u32 nq = F.counts[0], ns = F.counts[1];
  const u32 ns_r = (ns + 63u) & ~63u;
  const Box3 sbox = F.var->sbox;
  // Own cells: cell_adr ascends with the compact index, so the lowest claimer of a component inside
  // a 1024-cell chunk is simply its first active cell.  One atomic per chunk for the chunk's leading
  // component (a frontier surface is mostly ONE component: per-wave atomics on its claim word were
  // thousands of same-address operations), wave-aggregated atomics for the other components.
  __shared__ u32 s_root[16], s_adr[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 base = blockIdx.x * 1024u; base < nq; base += gridDim.x * 1024u) {
    bool act[4];
    u32 rt[4], ad[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 i = base + (u32)k * 256u + threadIdx.x;
      act[k] = false;
      rt[k] = 0u;
      ad[k] = 0u;
      if (i < nq) {
        ad[k] = F.cell_adr[i];
        // flatten on the way: every cell ends with parent = root (k_sizes / the seed claims read it);
        // a plain store is safe here, nothing issues atomics on parent[] in this kernel
        const u32 root = uf_root(F.parent, i);
        F.parent[i] = root;
        if (in_box(g, sbox, ad[k])) {
          act[k] = true;
          rt[k] = root;
        }
      }
      const u64 m = __ballot(act[k]);
      if (lane == 0) s_root[k * 4 + wave] = NOCLAIM;
      if (m && lane == __builtin_ctzll(m)) {
        s_root[k * 4 + wave] = rt[k];
        s_adr[k * 4 + wave] = ad[k];
      }
    }
    __syncthreads();
    u32 key0 = NOCLAIM, adr0 = 0u;
    for (int e = 15; e >= 0; --e)
      if (s_root[e] != NOCLAIM) {
        key0 = s_root[e];
        adr0 = s_adr[e];
      }
    if (threadIdx.x == 0 && key0 != NOCLAIM) atomicMin(&F.claim[key0], adr0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool slow = act[k] && rt[k] != key0;
      if (__ballot(slow)) wave_min_claim(F.claim, slow, rt[k], ad[k]);
    }
    __syncthreads();
  }
  // NQ seeds: claim every component touching the seed's 26-neighbourhood.  The nine z-lines around
  // the seed are read as 3-bit windows first (independent loads); only lines that hold Q0 cells
  // take the dependent rank -> root -> claim chain, once per run (101 holds two runs).
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < ns_r; i += gridDim.x * blockDim.x) {
    const bool live = i < ns;
    const long a = live ? F.seed_adr[i] : 0;
    u32 pats = 0u;
    if (live) {
      const int x = (int)(a / g.nyz);
      const int rr = (int)(a - (long)x * g.nyz);
      const int y = rr / g.nz, z = rr - y * g.nz;
#pragma unroll
      for (int l = 0; l < 9; ++l) {
        const int dx = l / 3 - 1, dy = l % 3 - 1;
        const int xx = x + dx, yy = y + dy;
        if (xx < 0 || xx >= g.nx || yy < 0 || yy >= g.ny) continue;
        u32 p = (u32)(plane_window(F.qb, a + (long)dx * g.nyz + (long)dy * g.nz - 1) & 7ull);
        if (z == 0) p &= ~1u;
        if (z == g.nz - 1) p &= ~4u;
        pats |= p << (3 * l);
      }
    }
    u32 last = NOCLAIM;
    while (__ballot(pats != 0u)) {
      bool active = pats != 0u;
      u32 r = 0u, r2 = 0u;
      bool two = false;
      if (active) {
        const int l = __builtin_ctz(pats) / 3;
        const u32 p = (pats >> (3 * l)) & 7u;
        pats &= ~(7u << (3 * l));
        const int dx = l / 3 - 1, dy = l % 3 - 1;
        const u32 j = rank_q(F, a + (long)dx * g.nyz + (long)dy * g.nz - 1 + __builtin_ctz(p));
        active = j < F.cap_q;
        if (active) {
          r = uf_root(F.parent, j);
          if (p == 5u && j + 1 < F.cap_q) {
            r2 = uf_root(F.parent, j + 1);
            two = r2 != r;
          }
          if (r == last) active = false;  // this seed already claimed that component
          last = r;
        }
      }
      if (__ballot(active)) wave_min_claim(F.claim, active, r, (u32)a);
      if (__ballot(two)) wave_min_claim(F.claim, two, r2, (u32)a);
    }
  }
}

// one atomicAdd per distinct key per wave
__device__ __forceinline__ void wave_agg_add(u32* base, bool active, u32 key) {
  u64 todo = __ballot(active);
  const int lane = threadIdx.x & 63;
  while (todo) {
    int leader = __builtin_ctzll(todo);
    u32 k = __shfl(key, leader, 64);
    u64 same = __ballot(active && key == k) & todo;
    if (lane == leader) atomicAdd(&base[k], (u32)__popcll(same));
    todo &= ~same;
  }
}

// cluster sizes by slot: own claimer -> its compact index; NQ seed claimer -> nq + seed rank.
// A frontier surface is typically ONE huge cluster, so per-lane (even per-wave) atomics on its
// counter serialise: every 1024-cell chunk is reduced in the block for its leading slot (key0) and
// only cells of other slots take the wave-aggregated atomic path.
__global__ void __launch_bounds__(256) k_sizes(Geo g, FArgs F) {
  __shared__ u32 s_key0;
  __shared__ u32 s_part[4];
  const u32 nq = F.counts[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 base = blockIdx.x * SZ_CH; base < nq; base += gridDim.x * SZ_CH) {
    u32 slots[SZ_CH / 256];
    bool act[SZ_CH / 256];
#pragma unroll
    for (int k = 0; k < SZ_CH / 256; ++k) {
      const u32 i = base + k * 256 + threadIdx.x;
      act[k] = false;
      slots[k] = 0;
      if (i < nq) {
        u32 cl = F.claim[F.parent[i]];
        if (cl != NOCLAIM) {
          act[k] = true;
          bool own = (F.qb[cl >> 6] >> (cl & 63)) & 1ull;
          slots[k] = own ? rank_q(F, cl) : nq + rank_s(F, cl);
          F.cell_slot[i] = (int)slots[k];
        } else
          F.cell_slot[i] = -1;
      }
    }
    if (wave == 0) {
      u64 m = __ballot(act[0]);
      u32 k0 = m ? (u32)__shfl((int)slots[0], __builtin_ctzll(m), 64) : NOCLAIM;
      if (lane == 0) s_key0 = k0;
    }
    __syncthreads();
    const u32 key0 = s_key0;
    u32 local = 0;
#pragma unroll
    for (int k = 0; k < SZ_CH / 256; ++k) {
      const bool fast = act[k] && slots[k] == key0;
      local += fast ? 1u : 0u;
      wave_agg_add(F.csize, act[k] && !fast, slots[k]);
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor((int)local, off, 64);
    if (lane == 0) s_part[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0 && key0 != NOCLAIM) atomicAdd(&F.csize[key0], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
    __syncthreads();
  }
}

// flags (all claimed cells + all NQ seeds), kept-cluster list, per-cell kept slot
__global__ void __launch_bounds__(256) k_finalize(Geo g, FArgs F) {
  if ((int)blockIdx.x >= F.var->nblocks) return;
  const int rel = blockIdx.x * 256 + threadIdx.x;
  const int w = F.var->w0 + rel;
  if (w >= g.W) return;
  u64 q = F.qb[w], s = F.sb[w];
  if ((q | s) == 0ull) return;
  const u32 nq = F.counts[0];
  u64 pk = F.blockscan[rel >> 8] + F.pref[rel];
  u32 iq = (u32)pk, is = (u32)(pk >> 32);
  u64 newflag = s;
  u64 qq = q;
  while (qq) {
    int b = __builtin_ctzll(qq);
    qq &= qq - 1;
    if (iq < F.cap_q) {
      int slot = F.cell_slot[iq];
      if (slot >= 0) {
        newflag |= 1ull << b;
        u32 sz = F.csize[slot];
        if ((int)sz > F.cluster_min) {
          if ((u32)slot == iq) {  // this cell is the own claimer of a kept cluster
            u32 k = atomicAdd(&F.counts[3], 1u);
            if (k < F.cap_kept) {
              F.kept[3 * k] = (u32)(64L * w + b);
              F.kept[3 * k + 1] = (u32)slot;
              F.kept[3 * k + 2] = sz;
            }
          }
        } else
          F.cell_slot[iq] = -1;
      }
    }
    ++iq;
  }
  u64 ss = s;
  while (ss) {
    int b = __builtin_ctzll(ss);
    ss &= ss - 1;
    if (is < F.cap_s) {
      u32 slot = nq + is;
      u32 sz = F.csize[slot];
      if ((int)sz > F.cluster_min) {
        u32 k = atomicAdd(&F.counts[3], 1u);
        if (k < F.cap_kept) {
          F.kept[3 * k] = (u32)(64L * w + b);
          F.kept[3 * k + 1] = slot;
          F.kept[3 * k + 2] = sz;
        }
      }
    }
    ++is;
  }
  F.flag[w] |= newflag;
}

// ---- grouping of kept cells by cluster: stable radix multisplit ---------------------------------
// Every size (nq, nkept, n_out, digits, chunk counts) is read from device memory, so the whole tail
// is enqueued without a host round trip; grids are launched at their upper bound and surplus
// blocks exit at once.  counts[]: [0]=nq [1]=ns [2]=overflow [3]=nkept [4]=scratch [5]=n_out.

// rank the kept clusters by claimer address (= the reference's creation order) and lay out the
// grouped cell array: rank_i = #{j : addr_j < addr_i}, off_i = sum of their cell counts.
__global__ void __launch_bounds__(256) k_rank_kept(FArgs F) {
  const u32 nq = F.counts[0];
  const u32 nkept = min(F.counts[3], F.cap_kept);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nkept; i += gridDim.x * blockDim.x) {
    const u32 ai = F.kept[3 * i], si = F.kept[3 * i + 1], zi = F.kept[3 * i + 2];
    u32 rank = 0, off = 0;
    for (u32 j = 0; j < nkept; ++j) {
      const u32 aj = F.kept[3 * j];
      if (aj < ai) {
        ++rank;
        off += F.kept[3 * j + 2] - (F.kept[3 * j + 1] >= nq ? 1u : 0u);  // an NQ seed is not a Q0 cell
      }
    }
    F.slot2rank[si] = (int)rank;
    KeptRec& r = F.krec[rank];
    r.addr = ai, r.slot = si, r.size = zi, r.off = off;
    r.sum[0] = r.sum[1] = r.sum[2] = 0ull;
    for (int k = 0; k < 3; ++k) r.box[k] = 0xFFFFFFFFu, r.box[3 + k] = 0u;
    if (rank == nkept - 1) F.counts[5] = off + zi - (si >= nq ? 1u : 0u);
  }
}
struct MsPass {
  bool on;
  u32 n;
  int nb, ndig, shift;
  const u32 *key, *val;
  u32 *key_out, *val_out;
};
__device__ __forceinline__ MsPass ms_pass(const FArgs& F, int pass) {
  MsPass p;
  const u32 nkept = min(F.counts[3], F.cap_kept);
  p.on = nkept > 0 && (pass == 0 || nkept > 256);
  p.n = pass == 0 ? F.counts[0] : F.counts[5];
  p.nb = (int)((p.n + MS_CH - 1) / MS_CH);
  p.ndig = pass == 0 ? (int)min(nkept, 256u) : (int)((nkept - 1) >> 8) + 1;
  p.shift = 8 * pass;
  p.key = F.ms_key[pass], p.val = F.ms_val[pass];
  p.key_out = F.ms_key[1 - pass], p.val_out = F.ms_val[1 - pass];
  return p;
}
// (key, value) of input item i: pass 0 of the clustering chain derives them from the per-cell slot
// (what a separate k_ms_keys launch used to materialise); otherwise they are read from the buffers
__device__ __forceinline__ u32 ms_key_at(const FArgs& F, const MsPass& P, int pass, u32 i) {
  if (pass == 0 && F.keys_from_slots) {
    const int s = F.cell_slot[i];
    return s >= 0 ? (u32)F.slot2rank[s] : NOKEY;
  }
  return P.key[i];
}
__device__ __forceinline__ u32 ms_val_at(const FArgs& F, const MsPass& P, int pass, u32 i) {
  return (pass == 0 && F.keys_from_slots) ? F.cell_adr[i] : P.val[i];
}
// in-place exclusive scan of the ndig*nb histogram entries (one block)
__device__ void ms_scan_tail(const FArgs& F, const MsPass& P) {
  __shared__ u32 part[1024];
  u32* v = F.ms_hist;
  const int T = blockDim.x;
  const int cnt = P.ndig * P.nb;
  const int per = (cnt + T - 1) / T;
  const int b0 = threadIdx.x * per, b1 = min(cnt, b0 + per);
  u32 s = 0;
  for (int b = b0; b < b1; ++b) s += v[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {  // Hillis-Steele over the per-thread partials
    const u32 x = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += x;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - s;
  for (int b = b0; b < b1; ++b) {
    u32 x = v[b];
    v[b] = run;
    run += x;
  }
}
// histogram of the current 8-bit digit per block, digit-major: hist[d * nb + block]
__global__ void __launch_bounds__(256) k_ms_hist(FArgs F, int pass) {
  __shared__ u32 h[256];
  const MsPass P = ms_pass(F, pass);
  if (!P.on) return;
  for (int chunk = blockIdx.x; chunk < P.nb; chunk += gridDim.x) {  // any grid size covers all chunks
    h[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = (u32)chunk * MS_CH;
    for (int k = 0; k < MS_CH / 256; ++k) {
      u32 i = base + k * 256 + threadIdx.x;
      if (i < P.n) {
        u32 kk = ms_key_at(F, P, pass, i);
        if (kk != NOKEY) atomicAdd(&h[(kk >> P.shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < P.ndig) F.ms_hist[threadIdx.x * P.nb + chunk] = h[threadIdx.x];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(1024) k_ms_scan(FArgs F, int pass) {
  const MsPass P = ms_pass(F, pass);
  if (P.on) ms_scan_tail(F, P);
}
// stable scatter: position = scanned[d][block] + (# earlier elements of this block with digit d)
__global__ void __launch_bounds__(256) k_ms_scatter(FArgs F, int pass) {
  __shared__ u32 running[256];
  __shared__ u32 wcnt[4][256];
  const MsPass P = ms_pass(F, pass);
  if (!P.on) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int chunk = blockIdx.x; chunk < P.nb; chunk += gridDim.x) {
  running[threadIdx.x] = (int)threadIdx.x < P.ndig ? F.ms_hist[threadIdx.x * P.nb + chunk] : 0u;
  const u32 base = (u32)chunk * MS_CH;
  for (int k = 0; k < MS_CH / 256; ++k) {
    for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const u32 i = base + k * 256 + threadIdx.x;
    u32 kk = (i < P.n) ? ms_key_at(F, P, pass, i) : NOKEY;
    const bool active = kk != NOKEY;
    const u32 d = (kk >> P.shift) & 255u;
    u32 lane_rank = 0;
    u64 todo = __ballot(active);
    while (todo) {
      int leader = __builtin_ctzll(todo);
      u32 dl = __shfl(d, leader, 64);
      u64 same = __ballot(active && d == dl) & todo;
      if (active && d == dl) lane_rank = (u32)__popcll(same & ((1ull << lane) - 1ull));
      if (lane == leader) wcnt[wave][dl] = (u32)__popcll(same);
      todo &= ~same;
    }
    __syncthreads();
    if (active) {
      u32 pos = running[d] + lane_rank;
      for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
      P.key_out[pos] = kk;
      P.val_out[pos] = ms_val_at(F, P, pass, i);
    }
    __syncthreads();
    running[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    __syncthreads();
  }
  }
}
// computeFrontierInfo (:374-390) accumulators per cluster: sum of voxel indices and index AABB.
// Input is grouped by cluster, so a 1024-cell chunk nearly always holds one key: reduce it in the
// block into a per-chunk record (folded on the host); cells of other keys (a chunk straddling a
// cluster boundary) are reduced per wave and added to the cluster record with atomics.
__device__ __forceinline__ void info_atomics(FArgs& F, u32 k, u32 sx, u32 sy, u32 sz, u32 nx_, u32 ny_, u32 nz_,
                                             u32 mx, u32 my, u32 mz) {
  KeptRec& r = F.krec[k];
  atomicAdd(&r.sum[0], (unsigned long long)sx);
  atomicAdd(&r.sum[1], (unsigned long long)sy);
  atomicAdd(&r.sum[2], (unsigned long long)sz);
  atomicMin(&r.box[0], nx_);
  atomicMin(&r.box[1], ny_);
  atomicMin(&r.box[2], nz_);
  atomicMax(&r.box[3], mx);
  atomicMax(&r.box[4], my);
  atomicMax(&r.box[5], mz);
}
// counts + cluster records -> pinned host memory (one block, after k_ms_info's atomics on the records)
__device__ void pack_tail(const FArgs& F) {
  const u32 nkept = min(F.counts[3], F.cap_kept);
  const u32 words = nkept * (u32)(sizeof(KeptRec) / 4);
  const u32* src = reinterpret_cast<const u32*>(F.krec);
  u32* dst = reinterpret_cast<u32*>(F.h_rec);
  for (u32 i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x < 16) F.h_counts[threadIdx.x] = F.counts[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_ms_info(Geo g, FArgs F) {
  __shared__ u32 s_red[4][9];
  const u32 nkept = min(F.counts[3], F.cap_kept);
  if (nkept == 0) return;
  const int fin = nkept <= 256 ? 1 : 0;  // buffer holding the grouped output
  const u32* key = F.ms_key[fin];
  const u32* val = F.ms_val[fin];
  const u32 n = F.counts[5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 base = blockIdx.x * SZ_CH; base < n; base += gridDim.x * SZ_CH) {
    const u32 key0 = key[base];
    u32 sx = 0, sy = 0, sz = 0, nx_ = 0xFFFFFFFFu, ny_ = 0xFFFFFFFFu, nz_ = 0xFFFFFFFFu, mx = 0, my = 0, mz = 0;
#pragma unroll
    for (int k = 0; k < SZ_CH / 256; ++k) {
      const u32 i = base + k * 256 + threadIdx.x;
      u32 kk = i < n ? key[i] : key0;
      // keys >= nkept can only appear when the radix-pass estimate was wrong and this buffer is not
      // the grouped one yet (the host then re-runs the grouping and this kernel): ignore them
      const bool in = i < n && kk < nkept;
      const u32 a = in ? val[i] : 0u;
      if (i < n) F.h_cells[i] = a;  // posted write over PCIe, 16 B.. 256 B per wave, coalesced
      const u32 x = a / (u32)g.nyz, r = a - x * (u32)g.nyz, y = r / (u32)g.nz, z = r - y * (u32)g.nz;
      if (in && kk == key0) {
        sx += x, sy += y, sz += z;
        nx_ = min(nx_, x), ny_ = min(ny_, y), nz_ = min(nz_, z);
        mx = max(mx, x), my = max(my, y), mz = max(mz, z);
      }
      u64 todo = __ballot(in && kk != key0);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const u32 kl = (u32)__shfl((int)kk, leader, 64);
        const bool mine = in && kk == kl;
        const u64 same = __ballot(mine) & todo;
        u32 tx = mine ? x : 0, ty = mine ? y : 0, tz = mine ? z : 0;
        u32 ax = mine ? x : 0xFFFFFFFFu, ay = mine ? y : 0xFFFFFFFFu, az = mine ? z : 0xFFFFFFFFu;
        u32 bx = tx, by = ty, bz = tz;
        for (int off = 32; off > 0; off >>= 1) {
          tx += __shfl_xor((int)tx, off, 64);
          ty += __shfl_xor((int)ty, off, 64);
          tz += __shfl_xor((int)tz, off, 64);
          ax = min(ax, (u32)__shfl_xor((int)ax, off, 64));
          ay = min(ay, (u32)__shfl_xor((int)ay, off, 64));
          az = min(az, (u32)__shfl_xor((int)az, off, 64));
          bx = max(bx, (u32)__shfl_xor((int)bx, off, 64));
          by = max(by, (u32)__shfl_xor((int)by, off, 64));
          bz = max(bz, (u32)__shfl_xor((int)bz, off, 64));
        }
        if (lane == leader) info_atomics(F, kl, tx, ty, tz, ax, ay, az, bx, by, bz);
        todo &= ~same;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      sx += __shfl_xor((int)sx, off, 64);
      sy += __shfl_xor((int)sy, off, 64);
      sz += __shfl_xor((int)sz, off, 64);
      nx_ = min(nx_, (u32)__shfl_xor((int)nx_, off, 64));
      ny_ = min(ny_, (u32)__shfl_xor((int)ny_, off, 64));
      nz_ = min(nz_, (u32)__shfl_xor((int)nz_, off, 64));
      mx = max(mx, (u32)__shfl_xor((int)mx, off, 64));
      my = max(my, (u32)__shfl_xor((int)my, off, 64));
      mz = max(mz, (u32)__shfl_xor((int)mz, off, 64));
    }
    if (lane == 0) {
      u32* q = s_red[wave];
      q[0] = sx, q[1] = sy, q[2] = sz, q[3] = nx_, q[4] = ny_, q[5] = nz_, q[6] = mx, q[7] = my, q[8] = mz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) {
        sx += s_red[w][0], sy += s_red[w][1], sz += s_red[w][2];
        nx_ = min(nx_, s_red[w][3]), ny_ = min(ny_, s_red[w][4]), nz_ = min(nz_, s_red[w][5]);
        mx = max(mx, s_red[w][6]), my = max(my, s_red[w][7]), mz = max(mz, s_red[w][8]);
      }
      u32* rec = F.h_part + (size_t)(base / SZ_CH) * 10;
      rec[0] = key0, rec[1] = sx, rec[2] = sy, rec[3] = sz;
      rec[4] = nx_, rec[5] = ny_, rec[6] = nz_, rec[7] = mx, rec[8] = my, rec[9] = mz;
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_pack(FArgs F) { pack_tail(F); }

__global__ void k_load_var(const FVar* __restrict__ h, FVar* __restrict__ d) {
  const int n = (int)(sizeof(FVar) / 4);
  if ((int)threadIdx.x < n) reinterpret_cast<u32*>(d)[threadIdx.x] = reinterpret_cast<const u32*>(h)[threadIdx.x];
}
__global__ void k_zero_words(u64* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0ull;
}
__global__ void k_expand_flag_bits(const u64* __restrict__ bits, long n, char* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (char)((bits[i >> 6] >> (i & 63)) & 1ull);
}

// =================================================================================================
// FAST PATH of the clustering chain ("tile-root resolve")
//
// The legacy chain above runs the cross-tile merge, the claims, the cluster sizes, the kept list and
// its ranking on the ~10^5 CELLS through device-scope atomics: five dependent kernels (k_union ..
// k_rank_kept, ~60 us) whose time is atomic latency.  Here a tile describes each of its local
// components by ONE record (k_tile_ccl: size, lowest claimer inside the scan box, index sums, index box,
// cells per x-row) and lists the component pairs that touch across tile faces (k_tile_cross); one
// workgroup then does all of the above on those few thousand records in its LDS (k_resolve).
//
// Round 3: no compaction anywhere, and no predicate pass in front.  The round-2 chain compacted the cells in address
// order first (predicate + in-block prefix, scan of the block sums, ordered compaction) and every later kernel found
// a cell's neighbours through rank look-ups in those tables (three dependent loads each); regrouping the cells by
// cluster took a histogram kernel and a scatter kernel.  Now four kernels:
//   k_tile_ccl  : a tile evaluates the frontier predicate for its own voxels straight from the occupancy planes,
//                 keeps the Q0 / seed bits of its 32-voxel segments in per-tile arrays (tq / ts), labels its cells run
//                 by run in LDS, writes one record per component and the component number of every cell BY VOXEL
//                 ADDRESS (vlab, one byte per voxel)
//   k_tile_cross: relations across tile faces + NQ seed claims, neighbours looked up in tq / vlab
//   k_resolve   : unions, clusters, ranks; additionally the (kept cluster x tile column) prefix matrix
//   k_tile_out  : flags + the grouped cell list: a cell's position = offset of its cluster + cells of the cluster
//                 in earlier tile columns (matrix) + in earlier x-rows / earlier tiles of its own column (per-row
//                 counts of the column's components) + earlier cells of its own row of the tile
// fuelmi_frontier_reset costs no kernel: the finder owns two flag planes and swaps to the zeroed one.
// Capacity limits (FR_* in frontier_internal.h) are those of pathological inputs (noise-like occupancy);
// when one is hit, or cluster_min < 1 (every NQ seed is then a cluster of its own), the search runs the
// legacy chain instead.  Results are identical (tests run both).
// =================================================================================================

// ---- tiles ----------------------------------------------------------------------------------------
struct TileGeo {
  int tx, ty, x0, y0, nxl, nyl, TX, TY, nseg, items;
};
// What the tile kernels behind k_tile_ccl need before they can issue their bulk loads -- the per-search block, the
// chain's overflow verdict, the tile's component count and id base -- fetched by different lanes in ONE round trip and
// handed round through LDS.  (Read field by field from global memory, with an early return in between, this was
// five dependent round trips at the head of every workgroup.)
struct TilePro {
  FVar V;
  u32 ovf, nroots, gbase, pad;
};
__device__ __forceinline__ void stage_tile_pro(const FArgs& F, TilePro* sp) {
  constexpr int NV = (int)(sizeof(FVar) / 4);
  static_assert(sizeof(FVar) % 4 == 0 && NV + 3 <= 64, "FVar does not fit one wave's lanes");
  const int t = threadIdx.x;
  // (the address is selected, not the load: one load instruction, one wait)
  const u32* src = reinterpret_cast<const u32*>(F.var) + (t < NV ? t : 0);
  if (t == NV) src = F.counts + 2;
  if (t == NV + 1) src = F.t_nroots + blockIdx.x;
  if (t == NV + 2) src = F.t_base + blockIdx.x;
  if (t < NV + 3) reinterpret_cast<u32*>(sp)[t] = *src;
  __syncthreads();
}
__device__ __forceinline__ TileGeo tile_geo(const Geo& g, const FVar& V, int t) {
  TileGeo T;
  T.TX = V.ftx, T.TY = V.fty;
  T.tx = t / V.nty_f, T.ty = t - T.tx * V.nty_f;
  T.x0 = V.px0 + T.tx * T.TX, T.y0 = V.py0 + T.ty * T.TY;
  T.nxl = min(T.TX, V.px1 - T.x0 + 1), T.nyl = min(T.TY, V.py1 - T.y0 + 1);
  T.nseg = (g.nz + 31) >> 5;
  T.items = T.TX * T.TY * T.nseg;
  return T;
}
__device__ __forceinline__ long tile_line_adr(const Geo& g, const TileGeo& T, int line) {  // address of voxel z = 0
  const int lx = line / T.TY, ly = line - lx * T.TY;
  return (long)(T.x0 + lx) * g.nyz + (long)(T.y0 + ly) * g.nz;
}

#define FT_PER 8  // segments a lane of the tile kernels handles at most (tile segments / workgroup size)
#define FT_TRIP 2  // segments of a lane whose plane windows are fetched side by side (4 measured no faster than 2: the fetch
                   // is bound by the number of load instructions x cache lines they touch, not by their latency)
// ---- the predicate inside the tile kernel ----------------------------------------------------------------
// Q0 and NQ-seed bits of the 32 voxels from address a = (x, y, 32 c): knownfree && isNeighborUnknown
// (frontier_finder.cpp:862-877) && flag == 0, cut to the Q region / the scan box -- what k_pred / f1_word compute
// per 64-voxel word, per segment here.  The seven plane windows are independent loads.
struct SegPred {
  u32 q, s;
};
// 32 bits of a plane from (signed) bit index `bit`, through 32-bit loads: the planes are little-endian u64 words, so
// bit b of the plane is bit (b & 31) of the 32-bit word b >> 5 -- half the bytes of plane_window per window
// (both words of a window in ONE 8-byte load at 4-byte alignment -- the tile kernels are bound by how many load
// instructions and cache-line look-ups their scattered windows cost, and two 4-byte loads look the same line up twice;
// the word behind the last one of a plane lies in its zeroed margin)
struct __attribute__((packed, aligned(4))) W2 {
  u32 a, b;
};
struct __attribute__((packed, aligned(4))) W3 {
  u32 a, b, c;
};
__device__ __forceinline__ u32 plane_window32(const u64* __restrict__ p, long bit) {
  const u32* q = reinterpret_cast<const u32*>(p);
  const long wi = bit >> 5;
  const int sh = (int)(bit & 31);
  const W2 w = *reinterpret_cast<const W2*>(q + wi);
  return (u32)((((u64)w.b << 32) | (u64)w.a) >> sh);
}
__device__ __forceinline__ SegPred seg_predicate(const Geo& g, const FVar& V, const FArgs& F, int x, int y, int c) {
  const long a = (long)x * g.nyz + (long)y * g.nz + 32 * c;
  const int zn = min(32, g.nz - 32 * c);
  u64 W;  // the unknown plane at z = 32 c - 1 .. 32 c + 32 (bit 0 <-> z = 32 c - 1): three 32-bit words
  {
    const u32* q = reinterpret_cast<const u32*>(F.unk);
    const long wi = (a - 1) >> 5;
    const int sh = (int)((a - 1) & 31);
    const W3 w = *reinterpret_cast<const W3*>(q + wi);
    const u64 w01 = (u64)w.a | ((u64)w.b << 32);
    W = (w01 >> sh) | (((u64)w.c << 1) << (63 - sh));  // (no branch on sh: a load inside a branch is waited for inside it)
  }
  // Every window is fetched unconditionally and masked afterwards: the planes carry zeroed margins of more than a
  // slab on both sides, so the lines "before" y = 0 and "after" y = ny - 1 are addressable (they are the neighbouring
  // slab's lines, or margin), and seven independent loads cost one round trip where seven guarded ones cost seven.
  const u32 occw = plane_window32(F.occ, a);
  const u32 ym_raw = plane_window32(F.unk, a - g.nz), yp_raw = plane_window32(F.unk, a + g.nz);
  const u32 xm = plane_window32(F.unk, a - g.nyz), xp = plane_window32(F.unk, a + g.nyz);  // (zero margins beyond the map)
  const u32 fl_raw = plane_window32(F.flag, a);
  const u32 ym = y > 0 ? ym_raw : 0u, yp = y < g.ny - 1 ? yp_raw : 0u;
  const u32 fl = V.fresh ? 0u : fl_raw;
  u32 down = (u32)W, up = (u32)(W >> 2);
  const u32 self_unk = (u32)(W >> 1);
  if (c == 0) down &= ~1u;                                   // z = 0 has no lower neighbour
  if (g.nz - 32 * c <= 32) up &= ~(1u << (zn - 1));          // z = nz - 1 (this segment's last voxel) has no upper one
  const u32 valid = zn < 32 ? (1u << zn) - 1u : 0xFFFFFFFFu;
  const u32 f1 = ~occw & ~self_unk & (down | up | ym | yp | xm | xp) & valid & ~fl;
  auto zmask = [&](const Box3& b) -> u32 {
    if (x < b.lo[0] || x > b.hi[0] || y < b.lo[1] || y > b.hi[1]) return 0u;
    const int lo = max(32 * c, b.lo[2]) - 32 * c, hi = min(32 * c + 31, b.hi[2]) - 32 * c;
    if (lo > hi) return 0u;
    return ((hi - lo + 1 >= 32) ? 0xFFFFFFFFu : ((1u << (hi - lo + 1)) - 1u)) << lo;
  };
  const u32 mq = zmask(V.qreg), ms = zmask(V.sbox);
  SegPred r;
  r.q = f1 & mq;
  r.s = f1 & ms & ~mq;
  return r;
}

// The tile's segments: predicate, Q0 bits + exclusive prefix + CCL labels and the seed bits in the LDS (the caller
// copies both bit arrays into the per-tile arrays tq / ts -- what the later kernels and the neighbouring tiles read:
// coalesced, and no 64-bit plane that tiles ending in the middle of a word would have to share).  FT_TRIP segments per
// lane and trip.
template <int NT>
__device__ __forceinline__ u32 tile_load_pred(const Geo& g, const TileGeo& T, const FVar& V, const FArgs& F, u32* segb,
                                              u32* segpre, u32* s_wsum, u32* lab, u32* segs, unsigned short* cseg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (T.items + NT - 1) / NT;  // <= FT_PER (checked on the host)
  const int it0 = threadIdx.x * per;
  u32 cnt = 0u;
  // (line, segment) of the lane's first item; the following ones by counting (a division by a run-time value is ~25
  // instructions, and this phase is bound by instruction issue)
  int c, lx, ly;
  {
    const int line = it0 / T.nseg;
    c = it0 - line * T.nseg;
    lx = line / T.TY;
    ly = line - lx * T.TY;
  }
#pragma nounroll
  for (int k0 = 0; k0 < per; k0 += FT_TRIP) {
    SegPred r[FT_TRIP];
#pragma unroll
    for (int h = 0; h < FT_TRIP; ++h) {
      r[h].q = r[h].s = 0u;
      const int it = it0 + k0 + h;
      if (k0 + h < per && it < T.items && lx < T.nxl && ly < T.nyl) r[h] = seg_predicate(g, V, F, T.x0 + lx, T.y0 + ly, c);
      if (++c == T.nseg) {
        c = 0;
        if (++ly == T.TY) ly = 0, ++lx;
      }
    }
#pragma unroll
    for (int h = 0; h < FT_TRIP; ++h) {
      const int it = it0 + k0 + h;
      if (k0 + h < per && it < T.items) {
        segb[it] = r[h].q, segs[it] = r[h].s;  // (to memory later: a store in front of a barrier is a round trip)
        cnt += (u32)__popc(r[h].q);
      }
    }
  }
  u32 v = cnt;
  for (int off = 1; off < 64; off <<= 1) {
    const u32 t = (u32)__shfl_up((int)v, off, 64);
    if (lane >= off) v += t;
  }
  if (lane == 63) s_wsum[wave] = v;
  __syncthreads();
  u32 woff = 0u, tot = 0u;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    if (k < wave) woff += s_wsum[k];
    tot += s_wsum[k];
  }
  u32 run = woff + v - cnt;
#pragma nounroll
  for (int k = 0; k < per; ++k) {
    if (it0 + k >= T.items) break;
    u32 rem = segb[it0 + k];  // (the lane re-reads what it just wrote)
    segpre[it0 + k] = run;
    if (tot <= FR_TCELL) {  // labels of the CCL: start of the cell's z-run inside its segment (runs are pre-joined)
      u32 l = run;
      while (rem) {
        const int s0 = __builtin_ctz(rem);
        const u32 inv = ~(rem >> s0);
        const int len = min(inv ? __builtin_ctz(inv) : 32, 32 - s0);
        rem &= ~((len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << s0);
        for (int q = 0; q < len; ++q) lab[l + (u32)q] = l, cseg[l + (u32)q] = (unsigned short)(it0 + k);  // (cell -> segment:
        l += (u32)len;                                                         // what a bisection of the prefix would find)
      }
    }
    run += (u32)__popc(segb[it0 + k]);
  }
  if (threadIdx.x == 0) segpre[T.items] = tot;
  __syncthreads();
  return tot;
}
// a tile's own bits from the arrays (k_tile_cross, k_tile_out): coalesced, no shifting
template <int NT, bool SCAN>
__device__ __forceinline__ u32 tile_load_arrays(const TileGeo& T, const FArgs& F, u32* segb, u32* segpre, u32* s_wsum,
                                                u32* segs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32* tq = F.tq + (size_t)blockIdx.x * T.items;
  const u32* ts = F.ts + (size_t)blockIdx.x * T.items;
  const int per = (T.items + NT - 1) / NT;
  const int it0 = threadIdx.x * per;
  u32 b[FT_PER], b2[FT_PER];
#pragma unroll
  for (int k = 0; k < FT_PER; ++k) {
    b[k] = b2[k] = 0u;
    if (k < per && it0 + k < T.items) b[k] = tq[it0 + k], b2[k] = ts[it0 + k];
  }
  u32 cnt = 0u;
#pragma unroll
  for (int k = 0; k < FT_PER; ++k)
    if (k < per && it0 + k < T.items) {
      segb[it0 + k] = b[k], segs[it0 + k] = b2[k];
      cnt += (u32)__popc(b[k]);
    }
  if (!SCAN) {
    __syncthreads();
    return 0u;
  }
  u32 v = cnt;
  for (int off = 1; off < 64; off <<= 1) {
    const u32 t = (u32)__shfl_up((int)v, off, 64);
    if (lane >= off) v += t;
  }
  if (lane == 63) s_wsum[wave] = v;
  __syncthreads();
  u32 woff = 0u, tot = 0u;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    if (k < wave) woff += s_wsum[k];
    tot += s_wsum[k];
  }
  u32 run = woff + v - cnt;
#pragma unroll
  for (int k = 0; k < FT_PER; ++k)
    if (k < per && it0 + k < T.items) {
      segpre[it0 + k] = run;
      run += (u32)__popc(b[k]);
    }
  if (threadIdx.x == 0) segpre[T.items] = tot;
  __syncthreads();
  return tot;
}
// first run of set bits of m (m != 0): start bit, length; removes it from m
__device__ __forceinline__ void pop_run32(u32& m, int& s, int& len) {
  s = __builtin_ctz(m);
  const u32 inv = ~(m >> s);
  len = inv ? __builtin_ctz(inv) : 32;
  len = min(len, 32 - s);
  m &= ~((len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << s);
}
__device__ __forceinline__ void pop_run64(u64& m, int& s, int& len) {
  s = __builtin_ctzll(m);
  const u64 inv = ~(m >> s);
  len = inv ? __builtin_ctzll(inv) : 64;
  len = min(len, 64 - s);
  m &= ~((len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << s);
}

// Tile CCL.  Labels are SPARSE (one per Q0 cell of the tile, addressed through the tile-local prefix of the
// per-segment popcounts), so a tile costs the same LDS whatever nz is.  The unit of work is the z-RUN: the cells
// of a run share a label from the start, a run looks at each of the four lower z-lines through one 34-bit window
// and joins every run it finds there -- a wall costs one union per line, not one per cell.
template <int NT>
__global__ void __launch_bounds__(NT) k_tile_ccl(Geo g, FArgs F, const FVar V) {
  // the chain's first kernel: the per-search arguments arrive as a KERNEL ARGUMENT (the graph node's parameters are
  // rewritten before every launch: no memory read -- least of all one over PCIe -- stands in front of the tile's
  // loads); workgroup 0 leaves the device copy the later kernels read
  if (blockIdx.x == 0 && threadIdx.x == 0) *F.var_w = V;
  if ((int)blockIdx.x >= V.ntiles_f) return;
  const TileGeo T = tile_geo(g, V, blockIdx.x);
  const Box3 sbox = V.sbox;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nz = g.nz, nseg = T.nseg, items = T.items, TY = T.TY;
  u32* lab = reinterpret_cast<u32*>(smem_raw);       // [FR_TCELL] union-find over the tile-local cell indices
  u32* segb = lab + FR_TCELL;                         // [items] Q0 bits of each 32-voxel segment
  u32* segpre = segb + items;                         // [items + 1]
  u32* acc = segpre + items + 1;                      // [FR_TROOT][8]: sy, sz, claim, ly, lz, hy, hz, -
  u32* rrow = acc + FR_TROOT * 8;                     // [FR_TROOT][FR_TXS] cells per x-row
  unsigned short* rootno = reinterpret_cast<unsigned short*>(rrow + FR_TROOT * FR_TXS);  // [FR_TCELL] at a root: its number
  unsigned char* rcell = reinterpret_cast<unsigned char*>(rootno + FR_TCELL);            // [FR_TCELL] component number per cell
  u32* segs = reinterpret_cast<u32*>(rcell + FR_TCELL);  // [items] NQ seed bits (on their way to the ts array)
  u32* plist = reinterpret_cast<u32*>(rootno);  // [FR_TPAIR] touching runs (cell << 16 | cell); shares the space of rootno + rcell, which are filled afterwards
  unsigned short* cseg = reinterpret_cast<unsigned short*>(acc);  // [FR_TCELL] segment of every cell, for the pair walk; shares the space of acc + rrow, which are initialised behind it
  static_assert((size_t)FR_TCELL * sizeof(unsigned short) <= (size_t)FR_TROOT * 8 * 4 + (size_t)FR_TROOT * FR_TXS * 4, "cell -> segment table does not fit");
  static_assert((size_t)FR_TPAIR * sizeof(u32) <= FR_TCELL * sizeof(unsigned short) + FR_TCELL, "pair list does not fit");
  __shared__ u32 s_wsum[NT / 64];
  __shared__ u32 s_nroots, s_base, s_flag;
  __shared__ u32 s_cnt[4];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_nroots = 0u, s_flag = 0u, s_cnt[0] = s_cnt[1] = s_cnt[2] = s_cnt[3] = 0u;
  FR_DBG_MARK(F, blockIdx.x, 0);
  const u32 total = tile_load_pred<NT>(g, T, V, F, segb, segpre, s_wsum, lab, segs, cseg);  // (uniform; predicate, prefix, labels)
  {  // the bit arrays of the tile, for the later kernels and the neighbours (in flight during the phases below)
    u32* tq = F.tq + (size_t)blockIdx.x * items;
    u32* ts = F.ts + (size_t)blockIdx.x * items;
    for (int it = threadIdx.x; it < items; it += NT) tq[it] = segb[it], ts[it] = segs[it];
  }
  FR_DBG_MARK(F, blockIdx.x, 1);
  if (total == 0u || total > FR_TCELL) {
    if (threadIdx.x == 0) {
      F.t_nroots[blockIdx.x] = 0u;
      F.t_base[blockIdx.x] = 0u;
      if (total > FR_TCELL) F.fctr[FCTR(9)] = 11u;  // (codes 11..17 name the capacity for FUELMI_FR_TIMING / debugging)
    }
    return;
  }
  FR_DBG_MARK(F, blockIdx.x, 2);
  // ---- components.  (1) One lane per CELL looks at the four lower z-lines around it through 3-bit windows of the LDS
  // bit arrays (walk_pairs below): every pair of touching runs (segment seams included) goes into an LDS list, a
  // cell's pairs collected in registers first (one list reservation per wave and trip).
  // (2) The list is then joined by HOOK + COMPRESS rounds (Shiloach-Vishkin style): both labels of a pair are read
  // side by side and the larger root takes the smaller with a non-returning atomicMin, then every cell walks to its
  // root; repeated until a round finds every pair joined.  Labels only ever decrease and parent < child always
  // holds, so there are no cycles; a hook that lost a race is simply repeated in the next round.
  // Why not a lock-free union-find inside the walk (rounds 2 and 3 tried it three ways): merging hundreds of runs
  // into one surface makes every lane retry against the same growing tree -- measured 7 us for one union per lane --
  // and the retry loops run under the divergence of the walk. ----
  __shared__ u32 s_np, s_chg, s_povf;
  if (threadIdx.x == 0) s_np = 0u, s_chg = 0u, s_povf = 0u;
  __syncthreads();
  auto walk_pairs = [&](auto&& fn) {  // fn(first cell of an own run, a cell of a touching run of a lower line / the seam)
    // One lane per CELL: its segment from the cell -> segment table, its bit inside the segment, then one 3-bit window
    // (z - 1, z, z + 1) per lower line.  A cell whose z-predecessor is a cell too (same run) shares that cell's
    // windows except for the voxel z + 1 of each lower line -- and that one only matters when it starts a new run
    // there (the line's voxel z is empty): a wall costs one look per cell and line, not one pair.
    for (u32 l = threadIdx.x; l < total; l += NT) {
      const int it = cseg[l];  // (written beside the labels: a bisection of the prefix here was 11 dependent LDS reads)
      const u32 bits = segb[it];
      u32 k = l - segpre[it], rem = bits;  // the k-th set bit
      while (k--) rem &= rem - 1u;
      const int zz = __builtin_ctz(rem);
      const int line = it / nseg, c = it - line * nseg, lx = line / TY, ly = line - lx * TY;
      const int z = 32 * c + zz;
      const bool seam = zz == 0 && c > 0 && (segb[it - 1] >> 31);
      const u32 own = seam ? l : lab[l];  // label node of the cell's run (its first cell inside the segment)
      // the cell's pairs are collected in registers (slot 0: the seam, slots 1 + 2 q, 2 + 2 q: lower line q) and handed
      // over together behind the loop, where the wave has reconverged: one list reservation per wave and trip instead
      // of one per (cell, line) among whatever lanes happened to sit in the same iteration
      u32 pk[9];
      u32 nv = 0u;
      pk[0] = (l << 16) | (l - 1u);
      if (seam) nv |= 1u;
      const bool has_prev = seam || (zz > 0 && ((bits >> (zz - 1)) & 1u));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[1 + 2 * q] = pk[2 + 2 * q] = 0u;
        const int nlx = lx + (q < 3 ? -1 : 0), nly = ly + (q < 3 ? q - 1 : -1);
        if (nlx < 0 || nly < 0 || nly >= TY) continue;
        const int nline = nlx * TY + nly;
        const int zlo = z - 1;
        const int s0 = max(zlo, 0) >> 5;
        const u64 w = (u64)segb[nline * nseg + s0] | ((s0 + 1 < nseg) ? ((u64)segb[nline * nseg + s0 + 1] << 32) : 0ull);
        u32 pat = (zlo >= 0) ? (u32)((w >> (zlo - 32 * s0)) & 7ull) : (u32)((w << 1) & 6ull);
        if (z + 1 >= nz) pat &= 3u;
        if (has_prev) pat = (pat & 6u) == 4u ? 4u : 0u;
        if (!pat) continue;
        const int zn = zlo + __builtin_ctz(pat);
        const int nit = nline * nseg + (zn >> 5);
        const u32 ln = segpre[nit] + (u32)__popc(segb[nit] & ((1u << (zn & 31)) - 1u));
        pk[1 + 2 * q] = (own << 16) | ln;
        nv |= 1u << (1 + 2 * q);
        if (pat == 5u) {  // the next cell of that line sits at zlo + 2
          pk[2 + 2 * q] = (own << 16) | (ln + 1u);
          nv |= 1u << (2 + 2 * q);
        }
      }
      fn(pk, nv);
    }
  };
  walk_pairs([&](const u32 (&pk)[9], u32 nv) {
    // (all lanes of the trip arrive here together) positions by a ballot scan of the pair counts, one reservation
    const u32 cnt = (u32)__popc(nv);
    u32 excl = 0u, tot = 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {  // cnt <= 9
      const unsigned long long bal = __ballot((cnt >> b) & 1u);
      excl += (u32)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)) << b;
      tot += (u32)__popcll(bal) << b;
    }
    if (tot == 0u) return;  // (uniform)
    const int leader = __builtin_ctzll(__ballot(1));
    u32 at = 0u;
    if (lane == leader) at = atomicAdd(&s_np, tot);
    at = (u32)__shfl((int)at, leader, 64) + excl;
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if ((nv >> k) & 1u) {
        if (at < FR_TPAIR)
          plist[at] = pk[k];
        else
          s_povf = 1u;
        ++at;
      }
  });
  __syncthreads();
  {
    volatile u32* vl = lab;
    const u32 np = min(s_np, (u32)FR_TPAIR);
    for (;;) {
      bool any = false;
      for (u32 p = threadIdx.x; p < np; p += NT) {
        const u32 pr = plist[p];
        const u32 ru = vl[pr >> 16], rv = vl[pr & 0xFFFFu];  // (roots: the labels are flat at the start of a round)
        if (ru != rv) {
          atomicMin(&lab[max(ru, rv)], min(ru, rv));
          any = true;
        }
      }
      if (any) s_chg = 1u;
      __syncthreads();
      const bool again = s_chg != 0u;
      __syncthreads();
      if (!again) break;
      if (threadIdx.x == 0) s_chg = 0u;
      for (u32 i = threadIdx.x; i < total; i += NT) {
        u32 r = vl[i];
        for (;;) {
          const u32 rr = vl[r];
          if (rr == r) break;
          r = rr;
        }
        vl[i] = r;
      }
      __syncthreads();
    }
  }
  if (s_povf) {  // more touching pairs than the list holds (a tile of single-voxel runs): the plain union-find walk
    walk_pairs([&](const u32 (&pk)[9], u32 nv) {
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if ((nv >> k) & 1u) lds_union_h(lab, pk[k] >> 16, pk[k] & 0xFFFFu);
    });
  }
  __syncthreads();
  // (the cell -> segment table is dead: its space becomes the accumulators of the record phase, two barriers ahead)
  for (int t = threadIdx.x; t < FR_TROOT * 8; t += NT) {
    const int k = t & 7;
    acc[t] = (k >= 2 && k <= 4) ? 0xFFFFFFFFu : 0u;  // claim / box minima start at +inf
  }
  for (int t = threadIdx.x; t < FR_TROOT * FR_TXS; t += NT) rrow[t] = 0u;
  FR_DBG_MARK(F, blockIdx.x, 3);
  // ---- roots: dense numbers ----
  for (u32 l = threadIdx.x; l < total; l += NT)
    if (lab[l] == l) {
      const u32 la = atomicAdd(&s_nroots, 1u);
      rootno[l] = (unsigned short)min(la, 0xFFFFu);
    }
  __syncthreads();
  const u32 nroots = s_nroots;
  if (nroots > FR_TROOT) {
    if (threadIdx.x == 0) {
      F.fctr[FCTR(9)] = 12u;
      F.t_nroots[blockIdx.x] = 0u;
      F.t_base[blockIdx.x] = 0u;
    }
    return;
  }
  // ---- ids of the components: one contiguous range per tile inside the XCD's part of the id space (the
  // returning atomic is issued here so that its latency hides behind the record loop) ----
  if (threadIdx.x == 0) {
    const u32 xcd = blockIdx.x & 7u;
    const u32 b = atomicAdd(&F.fctr[FCTR(xcd)], nroots);
    if (b + nroots > FR_RC8) {
      F.fctr[FCTR(9)] = 13u;
      s_flag = 1u;
    }
    s_base = xcd * FR_RC8 + b;
  }
  // ---- component number of every cell ----
  for (u32 l = threadIdx.x; l < total; l += NT) rcell[l] = (unsigned char)rootno[lds_find_h(lab, l)];
  __syncthreads();
  FR_DBG_MARK(F, blockIdx.x, 4);
  // ---- per-component records.  One lane per z-LINE adds up its runs in registers (a line's cells nearly always
  // belong to one component); a wave whose lanes all hold the same component reduces across the lanes first. ----
  for (int line0 = 0; line0 < T.TX * TY; line0 += NT) {  // (uniform trip count: the wave reduces below)
    const int line = line0 + threadIdx.x;
    const bool act = line < T.TX * TY;
    const int lx = line / TY, ly = line - lx * TY;
    const u32 x = (u32)(T.x0 + lx), y = (u32)(T.y0 + ly);
    const bool xy_in = (int)x >= sbox.lo[0] && (int)x <= sbox.hi[0] && (int)y >= sbox.lo[1] && (int)y <= sbox.hi[1];
    const long la = (long)x * g.nyz + (long)y * nz;
    u32 key = 0xFFFFFFFFu, n = 0u, sz = 0u, cl = 0xFFFFFFFFu, lz = 0xFFFFFFFFu, hz = 0u;
    auto flush = [&]() {  // direct LDS atomics (the slow path: a second component on the line)
      u32* r = acc + key * 8u;
      atomicAdd(&r[0], n * y), atomicAdd(&r[1], sz), atomicMin(&r[2], cl), atomicMin(&r[3], y), atomicMin(&r[4], lz);
      atomicMax(&r[5], y), atomicMax(&r[6], hz);
      atomicAdd(&rrow[key * FR_TXS + (u32)lx], n);
    };
    for (int c = 0; act && c < nseg; ++c) {
      const int it = line * nseg + c;
      u32 rem = segb[it];
      u32 l = segpre[it];
      while (rem) {
        int s, len;
        pop_run32(rem, s, len);
        const u32 k2 = (u32)rcell[l];
        if (k2 != key) {
          if (key != 0xFFFFFFFFu) flush();
          key = k2, n = 0u, sz = 0u, cl = 0xFFFFFFFFu, lz = 0xFFFFFFFFu, hz = 0u;
        }
        const u32 z0 = (u32)(32 * c + s), z1 = z0 + (u32)len - 1u;
        n += (u32)len;
        sz += (u32)len * z0 + (u32)(len * (len - 1) / 2);
        lz = min(lz, z0), hz = max(hz, z1);
        if (xy_in) {
          const int zc = max((int)z0, sbox.lo[2]);
          if (zc <= min((int)z1, sbox.hi[2])) cl = min(cl, (u32)(la + zc));
        }
        l += (u32)len;
      }
    }
    // the last (usually only) component of the line: wave-uniform -> one set of atomics per wave
    const bool have = key != 0xFFFFFFFFu;
    const u64 hm = __ballot(have);
    if (hm) {
      const u32 first = (u32)__shfl((int)key, __builtin_ctzll(hm), 64);
      const bool uni = __ballot(have && key != first) == 0ull;
      if (uni) {
        u32 sy = have ? n * y : 0u, s2 = have ? sz : 0u, c2 = have ? cl : 0xFFFFFFFFu;
        u32 ly2 = have ? y : 0xFFFFFFFFu, lz2 = have ? lz : 0xFFFFFFFFu, hy2 = have ? y : 0u, hz2 = have ? hz : 0u;
        sy = wave_add_u32(sy), s2 = wave_add_u32(s2), c2 = wave_min_u32(c2), ly2 = wave_min_u32(ly2);
        lz2 = wave_min_u32(lz2), hy2 = wave_max_u32(hy2), hz2 = wave_max_u32(hz2);
        if (lane == __builtin_ctzll(hm)) {
          u32* r = acc + first * 8u;
          atomicAdd(&r[0], sy), atomicAdd(&r[1], s2), atomicMin(&r[2], c2), atomicMin(&r[3], ly2), atomicMin(&r[4], lz2);
          atomicMax(&r[5], hy2), atomicMax(&r[6], hz2);
        }
        if (have) atomicAdd(&rrow[key * FR_TXS + (u32)lx], n);
      } else if (have)
        flush();
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, blockIdx.x, 5);
  const u32 gbase = s_base;
  if (threadIdx.x == 0) {
    F.t_base[blockIdx.x] = gbase;
    F.t_nroots[blockIdx.x] = s_flag ? 0u : nroots;
  }
  if (s_flag) return;
  if (threadIdx.x < nroots) {
    const u32* r = acc + threadIdx.x * 8;
    const u32* rw = rrow + threadIdx.x * FR_TXS;
    TRec R;
    u32 size = 0u, sx = 0u, lox = 0xFFFFFFFFu, hix = 0u;
    unsigned short* grow = F.rrow + (size_t)(gbase + threadIdx.x) * FR_TXS;
    for (int lx = 0; lx < FR_TXS; ++lx) {
      const u32 cnt = lx < T.TX ? rw[lx] : 0u;
      grow[lx] = (unsigned short)cnt;  // (<= TY * nz <= 32 * 256 cells)
      if (cnt) {
        size += cnt, sx += cnt * (u32)(T.x0 + lx);
        lox = min(lox, (u32)(T.x0 + lx)), hix = max(hix, (u32)(T.x0 + lx));
      }
    }
    R.size = size, R.sx = sx, R.sy = r[0], R.sz = r[1];
    R.lo[0] = lox, R.lo[1] = r[3], R.lo[2] = r[4], R.hi[0] = hix, R.hi[1] = r[5], R.hi[2] = r[6];
    R.tx = (u32)T.tx, R.own = r[2];
    F.trec[gbase + threadIdx.x] = R;
    F.tclaim[gbase + threadIdx.x] = r[2];
  }
  // ---- component number of every cell, by tile-local index (k_tile_out) and by voxel address (k_tile_cross) ----
  {
    unsigned char* tl = F.tlab + (size_t)blockIdx.x * FR_TCELL;
    for (u32 l = threadIdx.x; l < total; l += NT) tl[l] = rcell[l];
  }
  for (int it = threadIdx.x; it < items; it += NT) {
    u32 rem = segb[it];
    if (!rem) continue;
    const int line = it / nseg, c = it - line * nseg;
    unsigned char* vl = F.vlab + tile_line_adr(g, T, line) + 32 * c;
    u32 l = segpre[it];
    while (rem) {
      const int b = __builtin_ctz(rem);
      rem &= rem - 1u;
      vl[b] = rcell[l++];
    }
  }
  FR_DBG_MARK(F, blockIdx.x, 6);
  if (F.dbg && threadIdx.x == 0) {
    F.dbg[(size_t)blockIdx.x * FR_DBG_SLOTS + 8] = total;
    F.dbg[(size_t)blockIdx.x * FR_DBG_SLOTS + 9] = s_cnt[0];
    F.dbg[(size_t)blockIdx.x * FR_DBG_SLOTS + 10] = s_cnt[1];
    F.dbg[(size_t)blockIdx.x * FR_DBG_SLOTS + 11] = s_cnt[2];
    F.dbg[(size_t)blockIdx.x * FR_DBG_SLOTS + 12] = nroots;
  }
}

// Joins across tile faces + seed claims, one launch after k_tile_ccl (every cell's component number is in vlab by
// now).  The runs of the tile's lower-face lines look across the face through the same 34-bit windows as inside
// the tile; the relation is recorded as a pair of TILE ROOTS, and a tile keeps one record per distinct pair (LDS
// set: a surface crossing a face gives the same pair from all of its cells) -- a few thousand records per search.
// The tile's NQ seeds claim the tile roots touching their 26-neighbourhood (atomicMin on tclaim).
#define XC_SET 256
#define XC_WCAP 6144  // entries of a tile's work list (beyond: processed on the spot)
template <int RS_T, bool IN_LAUNCH>
__device__ __forceinline__ bool resolve_body(const Geo& g, const FArgs& F, const FVar& V, unsigned char* smem_raw, const u32 rcap);

__device__ __forceinline__ void xc_insert(u32* s_set, u32* s_list, u32* s_n, u32* fctr, u32 key) {
  u32 h = (key * 2654435761u) >> 24;
  bool done = false;
  for (int probe = 0; probe < XC_SET && !done; ++probe) {
    const u32 old = atomicCAS(&s_set[h], 0xFFFFFFFFu, key);
    if (old == 0xFFFFFFFFu) {
      s_list[atomicAdd(s_n, 1u)] = key;
      done = true;
    } else if (old == key)
      done = true;
    h = (h + 1u) & (XC_SET - 1u);
  }
  if (!done) st_agent(&fctr[FCTR(9)], 15u);  // more than XC_SET distinct root pairs around one tile
}
// rcap != 0 (round 6): the workgroup that finishes LAST joins the tile roots itself (resolve_body, in the LDS its tile no
// longer needs) -- k_resolve's launch and kernel boundary leave the search's critical path.  What it reads of the
// other workgroups of this launch (pairs, claims, overflow codes) was written with agent-scope stores / memory-side
// atomics and is counted in behind s_waitcnt vmcnt(0): no release fence, no L2 write-back.  rcap = tile roots that fit
// the launch's LDS; searches with more leave the job to the kernel k_resolve behind (which returns at once otherwise).
template <int NT>
__global__ void __launch_bounds__(NT) k_tile_cross(Geo g, FArgs F, const u32 rcap) {
  FR_DBG_MARK(F, blockIdx.x, 13);  // (before the first load)
  __shared__ TilePro s_pro;
  stage_tile_pro(F, &s_pro);
  const FVar& V = s_pro.V;
  if ((int)blockIdx.x >= V.ntiles_f) return;
  // (no look at the chain's overflow word here: it shares a cache line with the counters k_tile_ccl just hit with
  // atomics, and thousands of waves waiting for that line cost more than the kernel.  After an overflow the
  // component numbers of some tile are stale: ids stay inside the padded tables, k_resolve discards everything.)
  const TileGeo T = tile_geo(g, V, blockIdx.x);
  const int nz = g.nz, nseg = T.nseg, TY = T.TY;
  const int lane = threadIdx.x & 63;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32* segq = reinterpret_cast<u32*>(smem_raw);  // [items] Q0 bits of the tile
  u32* segs = segq + T.items;                    // [items] NQ seed bits
  // work list: (own segment, neighbour line) combinations that have something to look up -- built from LDS only,
  // so that the global loads behind them (neighbour window, then component numbers) go out together.  An entry is
  // the item's number: face item fs * 4 + k, or seed item 0x80000000 | it * 9 + l.
  u32* wl = segs + T.items;                      // [XC_WCAP]
  __shared__ u32 s_set[XC_SET];   // distinct (root, root) pairs of this tile: open addressing
  __shared__ u32 s_list[XC_SET];  // ... in insertion order
  __shared__ u32 s_n, s_base, s_nw;
  __shared__ u32 s_tb[9];         // component-id base of the 3 x 3 tiles around this one
  const int dblk = V.ntiles_f + 1 + (int)blockIdx.x;
  FR_DBG_MARK(F, dblk, 0);
  u32 tb9;  // id base of one of the 3 x 3 tiles around this one (in flight beside the tile's arrays)
  {
    const int t9 = (int)threadIdx.x % 9, dtx = t9 / 3 - 1, dty = t9 % 3 - 1;
    const int ntx = T.tx + dtx, nty = T.ty + dty;
    const bool ok = ntx >= 0 && ntx < V.ntx_f && nty >= 0 && nty < V.nty_f;
    tb9 = F.t_base[ok ? ntx * V.nty_f + nty : (int)blockIdx.x];
    if (!ok) tb9 = 0u;
  }
  if (threadIdx.x < XC_SET) s_set[threadIdx.x] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) s_n = 0u, s_nw = 0u;
  tile_load_arrays<NT, false>(T, F, segq, nullptr, nullptr, segs);
  if (threadIdx.x < 9) s_tb[threadIdx.x] = tb9;  // (read behind the barrier that ends the work-list phase)
  FR_DBG_MARK(F, dblk, 1);
  auto tile_of = [&](int x, int y) -> u32 {  // which of the 3 x 3 tiles holds column (x, y)
    const int dtx = x < T.x0 ? -1 : (x >= T.x0 + T.TX ? 1 : 0), dty = y < T.y0 ? -1 : (y >= T.y0 + TY ? 1 : 0);
    return (u32)((dtx + 1) * 3 + dty + 1);
  };
  // one item: the neighbour window w3 (bit j <-> z = 32 c - 1 + j of the neighbour line) against the own bits
  u32 k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;  // up to two distinct pairs wait for the wave-level de-duplication
  auto do_item = [&](u32 bits, u32 own, int nb, u32 info, u64 w3) {
    const int c = (int)(info & 0xFFu);
    const u32 tb = s_tb[(info >> 8) & 0xFFu];
    w3 &= (1ull << 34) - 1ull;
    if (c == 0) w3 &= ~1ull;
    const int jend = nz - 32 * c + 1;  // bit position of z = nz
    if (jend < 34) w3 &= (1ull << jend) - 1ull;
    if (info >> 16) {
      // seeds: a component touching the segment's seeds is claimed by the lowest seed next to one of its cells
      while (w3) {
        int j, rl;
        pop_run64(w3, j, rl);
        const int b0 = max(j - 2, 0), b1 = min(j + rl - 1, 31);  // seeds at bits j - 2 .. j + rl - 1 touch the run
        if (b0 > b1) continue;
        const u32 m = bits & (((b1 - b0 + 1 >= 32) ? 0xFFFFFFFFu : ((1u << (b1 - b0 + 1)) - 1u)) << b0);
        if (!m) continue;
        const u32 r = tb + (u32)F.vlab[(long)nb + j];
        atomicMin(&F.tclaim[r], own + (u32)__builtin_ctz(m));
      }
      return;
    }
    u32 rem = bits;
    while (rem && w3) {
      int s, len;
      pop_run32(rem, s, len);
      u64 m = (w3 >> s) & ((1ull << (len + 2)) - 1ull);
      if (!m) continue;
      const u32 ga = s_tb[4] + (u32)F.vlab[(long)own + s];
      while (m) {
        int j, rl;
        pop_run64(m, j, rl);
        const u32 gb = tb + (u32)F.vlab[(long)nb + s + j];
        if (gb == ga) continue;
        const u32 key = (min(ga, gb) << 16) | max(ga, gb);
        if (key == k0 || key == k1) continue;
        if (k0 == 0xFFFFFFFFu)
          k0 = key;
        else if (k1 == 0xFFFFFFFFu)
          k1 = key;
        else
          xc_insert(s_set, s_list, &s_n, F.fctr, key);  // (a third distinct pair from one lane: rare)
      }
    }
  };
  auto flush_pairs = [&]() {  // (all lanes of the wave)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const u32 kq = q ? k1 : k0;
      const bool on = kq != 0xFFFFFFFFu;
      u64 todo = __ballot(on);
      while (todo) {  // one insertion per distinct pair of the wave
        const int leader = __builtin_ctzll(todo);
        const u32 key = (u32)__shfl((int)kq, leader, 64);
        todo &= ~__ballot(on && kq == key);
        if (lane == leader) xc_insert(s_set, s_list, &s_n, F.fctr, key);
      }
    }
    k0 = k1 = 0xFFFFFFFFu;
  };
  // what an entry stands for: own bits, address of the own segment, address of the neighbour window, info word
  auto decode = [&](u32 e, u32& bits, u32& own, int& nb, u32& info, int& xx, int& yy) {
    int it, c;
    if (e >> 31) {
      const int sj = (int)(e & 0x7FFFFFFFu), l = sj % 9;
      it = sj / 9;
      const int line = it / nseg, lx = line / TY, ly = line - lx * TY;
      c = it - line * nseg;
      xx = T.x0 + lx + l / 3 - 1, yy = T.y0 + ly + l % 3 - 1;
      bits = (xx >= V.px0 && xx <= V.px1 && yy >= V.py0 && yy <= V.py1) ? segs[it] : 0u;  // (no Q0 cells outside the tiles)
      own = (u32)((long)(T.x0 + lx) * g.nyz + (long)(T.y0 + ly) * nz + 32 * c);
      info = (u32)c | (tile_of(xx, yy) << 8) | (1u << 16);
    } else {
      const int k = (int)(e & 3u), fs = (int)(e >> 2), fl = fs / nseg;
      c = fs - fl * nseg;
      int lx = 0, ly = fl;
      if (fl >= TY) lx = 1 + ((fl - TY) >> 1), ly = ((fl - TY) & 1) ? TY - 1 : 0;
      xx = T.x0 + lx + (k < 3 ? -1 : 0), yy = T.y0 + ly + (k < 3 ? k - 1 : -1);
      bits = segq[(lx * TY + ly) * nseg + c];
      own = (u32)((long)(T.x0 + lx) * g.nyz + (long)(T.y0 + ly) * nz + 32 * c);
      info = (u32)c | (tile_of(xx, yy) << 8);
    }
    nb = (int)((long)xx * g.nyz + (long)yy * nz + 32 * c - 1);
  };
  // ---- tile faces: the whole x-row 0 and, in the other x-rows, the lines ly = 0 and ly = TY - 1; an item per
  // (face line, segment, lower z-line across the face) ----
  const int nfl = TY + 2 * (T.nxl - 1);
  const int fitems = nfl * nseg * 4;
  for (int fi = threadIdx.x; fi < fitems; fi += NT) {
    const int k = fi & 3, fs = fi >> 2, fl = fs / nseg, c = fs - fl * nseg;
    int lx = 0, ly = fl;
    if (fl >= TY) lx = 1 + ((fl - TY) >> 1), ly = ((fl - TY) & 1) ? TY - 1 : 0;
    if (!segq[(lx * TY + ly) * nseg + c]) continue;
    const int kdx = k < 3 ? -1 : 0, kdy = k < 3 ? k - 1 : -1;
    const int xx = T.x0 + lx + kdx, yy = T.y0 + ly + kdy;
    const bool cross = (kdx < 0 && lx == 0) || (kdy < 0 && ly == 0) || (kdy > 0 && ly == TY - 1);
    if (!(cross && ly < T.nyl && xx >= V.px0 && yy >= V.py0 && yy <= V.py1)) continue;
    const u32 slot = atomicAdd(&s_nw, 1u);
    if (slot < XC_WCAP)
      wl[slot] = (u32)fi;
    else
      st_agent(&F.fctr[FCTR(9)], 18u);  // work list full (a tile made of seeds): the legacy chain takes the search
  }
  // ---- NQ seeds of the tile (most tiles have none): nine items per seed segment, one per line around it (one
  // reservation for all nine) ----
  for (int it = threadIdx.x; it < T.items; it += NT) {
    if (!segs[it]) continue;
    const u32 slot0 = atomicAdd(&s_nw, 9u);
    for (int l = 0; l < 9; ++l) {
      const u32 e = 0x80000000u | (u32)(it * 9 + l);
      if (slot0 + (u32)l < XC_WCAP)
        wl[slot0 + (u32)l] = e;
      else
        st_agent(&F.fctr[FCTR(9)], 18u);
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 2);
  // ---- the list: two items per lane and trip, their windows fetched together ----
  const u32 nw = min(s_nw, (u32)XC_WCAP);
  for (u32 w0 = 0; w0 < nw; w0 += 2 * NT) {
    u32 b0 = 0u, b1 = 0u, o0 = 0u, o1 = 0u, i0 = 0u, i1 = 0u;
    int n0 = 0, n1 = 0, xa = 0, ya = 0, xb = 0, yb = 0;
    u64 wa = 0ull, wb = 0ull;
    const u32 ja = w0 + threadIdx.x, jb = ja + NT;
    if (ja < nw) decode(wl[ja], b0, o0, n0, i0, xa, ya);
    if (jb < nw) decode(wl[jb], b1, o1, n1, i1, xb, yb);
    // (both windows unconditionally -- an idle slot looks at column (0, 0), outside the rectangle or not -- so that
    // the six loads go out together)
    wa = q_window34(g, V, F, xa, ya, (int)(i0 & 0xFFu));
    wb = q_window34(g, V, F, xb, yb, (int)(i1 & 0xFFu));
    if (!b0) wa = 0ull;
    if (!b1) wb = 0ull;
#pragma nounroll
    for (int h = 0; h < 2; ++h) {  // (one copy of the item code)
      const u32 bb = h ? b1 : b0;
      const u64 ww = h ? wb : wa;
      if (bb && ww) do_item(bb, h ? o1 : o0, h ? n1 : n0, h ? i1 : i0, ww);
      flush_pairs();
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 3);
  const u32 n = s_n;
  if (n != 0u) {  // (uniform)
    const u32 xcd = blockIdx.x & 7u;
    if (threadIdx.x == 0) s_base = atomicAdd(&F.fctr[FCTR(16 + xcd)], n);
    __syncthreads();
    const u32 base = s_base;
    if (base + n > FR_PCAP / 8u) {
      if (threadIdx.x == 0) st_agent(&F.fctr[FCTR(9)], 14u);
    } else if (threadIdx.x < n)
      st_agent(&F.pairs[(size_t)xcd * (FR_PCAP / 8u) + base + threadIdx.x], s_list[threadIdx.x]);
  }
  FR_DBG_MARK(F, dblk, 4);
  if constexpr (NT == 512) {
    if (rcap == 0u) return;
    // ---- last one out joins the roots ----
    __shared__ u32 s_last;
    wait_vm_stores();  // (every wave: its pairs and codes have left)
    __syncthreads();
    // (the counter has a cache line of its own: on the line of the pair-list counters its 576 returning atomics stood in
    // front of every tile's list reservation -- the busiest tile's cross phase 13 -> 19 us)
    // two levels: the workgroups of an XCD (blockIdx & 7) count into that XCD's word, the last of each into the global one:
    // 72 + 8 returning atomics per line instead of 576 on one
    if (threadIdx.x == 0) {
      const u32 xcd = blockIdx.x & 7u, n_x = ((u32)V.ntiles_f + 7u - xcd) >> 3;  // workgroups b < ntiles_f with b & 7 == xcd
      u32 last = 0u;
      if (atomicAdd(&F.fctr[FR_DONE_CTR + FCTR(1 + xcd)], 1u) == n_x - 1u) {
        const u32 n_groups = min((u32)V.ntiles_f, 8u);
        last = atomicAdd(&F.fctr[FR_DONE_CTR], 1u) == n_groups - 1u ? 1u : 0u;
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const bool no_kr = (rcap >> 31) != 0u;  // no k_resolve is queued behind this launch (the host expected this search to fit)
    if (resolve_body<NT, true>(g, F, V, smem_raw, rcap & 0x7FFFFFFFu) || !no_kr) return;
    // more tile roots than this launch's LDS holds and nobody behind to do the job: tell the host (it queues k_resolve and
    // k_tile_out again and waits for the stamp a second time) and make the k_tile_out already queued return at once
    if (threadIdx.x == 0) {
      F.counts[2] = 3u;
      F.h_counts[2] = 3u;
      F.h_counts[6] = 20u;
      __hip_atomic_store(&F.h_counts[15], V.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// one wave-level reduction step of the per-component accumulators (sum / min / max over the lanes that share
// the leading key)
struct CAcc {
  u32 n, sx, sy, sz, cl, lx, ly, lz, hx, hy, hz;
};
__device__ __forceinline__ void cacc_reduce(CAcc& a) {
  a.n = wave_add_u32(a.n), a.sx = wave_add_u32(a.sx), a.sy = wave_add_u32(a.sy), a.sz = wave_add_u32(a.sz);
  a.cl = wave_min_u32(a.cl), a.lx = wave_min_u32(a.lx), a.ly = wave_min_u32(a.ly), a.lz = wave_min_u32(a.lz);
  a.hx = wave_max_u32(a.hx), a.hy = wave_max_u32(a.hy), a.hz = wave_max_u32(a.hz);
}

// Everything between "tile-local components" and "kept clusters in creation order", on the tile-root records,
// inside one workgroup: cross-tile union-find, sizes and claims per final component, clusters (a component
// claimed by one of its own cells is a cluster; components claimed by the same NQ seed form one with it), the
// kept list, its ranking by claimer address (= the reference's creation order), offsets of the grouped cell
// array, per-cluster index sums / boxes, the (cluster x tile column) prefix matrix k_tile_out places the cells
// with -- and the result records, written straight to pinned host memory.
#define RS_TK 1024  // lanes of the kernel k_resolve
#define RS_SH 512   // seed-claimed clusters (hash slots)
// bytes of LDS resolve_body needs for `rcap` tile roots
static inline size_t resolve_lds_bytes(size_t rcap) {
  return (4 * rcap + 6 * (size_t)FR_KCAP + 6 * (size_t)FR_KCAP + 3 * (size_t)RS_SH) * sizeof(u32) + 3 * (size_t)FR_KCAP * sizeof(unsigned long long);
}
// RS_T lanes of ONE workgroup; rcap = tile roots the LDS arrays hold (FR_RCAP in the kernel k_resolve; what the LDS of
// k_tile_cross holds when its last workgroup does the job).  IN_LAUNCH: pairs, claims and overflow codes were written
// by other workgroups of the SAME launch (agent-scope stores / memory-side atomics): they are read with agent-scope
// loads, past this XCD's L2 (the records are the previous kernel's and would be visible anyway; they take the same
// path).  Returns false when the search has more tile roots than rcap (nothing written: the kernel k_resolve, always
// queued behind, does the work).
template <int RS_T, bool IN_LAUNCH>
__device__ __forceinline__ bool resolve_body(const Geo& g, const FArgs& F, const FVar& V, unsigned char* smem_raw, const u32 rcap) {
  auto ldw = [&](const u32* p) -> u32 { return IN_LAUNCH ? ld_agent(p) : *p; };
  u32* par = reinterpret_cast<u32*>(smem_raw);  // [rcap] union-find over the DENSE numbers of the tile roots
  u32* siz = par + rcap;                        // [rcap] cells of the final component (at its root); later the matrix
  u32* clm = siz + rcap;                        // [rcap] lowest claimer address (at its root); later the matrix
  u32* rko = clm + rcap;                        // [rcap] lowest OWN claimer (at its root), then: index into the kept list | FR_NOTKEPT | FR_UNCLAIMED
  u32* k_adr = rko + rcap;                      // [FR_KCAP] kept list: claimer address
  u32* k_slot = k_adr + FR_KCAP;                //   0: claimed by an own cell, 0xFFFFFFFF: by an NQ seed
  u32* k_size = k_slot + FR_KCAP;               //   cluster size (the seed counts)
  u32* k_nq = k_size + FR_KCAP;                 //   its Q0 cells
  u32* k_rank = k_nq + FR_KCAP;                 //   rank by claimer address
  u32* k_off = k_rank + FR_KCAP;                //   first position in the grouped cell array
  u32* kbox = k_off + FR_KCAP;                  // [FR_KCAP][6] index box per kept cluster (by kept index)
  u32* h_adr = kbox + FR_KCAP * 6;              // [RS_SH] seed hash: claimer address
  u32* h_sum = h_adr + RS_SH;                   //   cells of the components it claims
  u32* h_kept = h_sum + RS_SH;                  //   index into the kept list | FR_NOTKEPT
  unsigned long long* ksum = reinterpret_cast<unsigned long long*>(h_kept + RS_SH);  // [FR_KCAP][3] index sums
  u32* pmx = siz;                               // [nk][ntx] (2 rcap entries) once siz / clm are dead
  __shared__ u32 s_nk, s_ovf, s_nout, s_nq;
  __shared__ u32 s_fc[32];  // the chain's counters (one round trip for all of them)
  __shared__ u32 s_pre[9];  // dense number of the first tile root of every XCD range
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntx = V.ntx_f;
  if (threadIdx.x < 32) s_fc[threadIdx.x] = ldw(&F.fctr[FCTR(threadIdx.x)]);
  const int dblk = V.ntiles_f;  // time stamps of this kernel go behind those of the tiles
  FR_DBG_MARK(F, dblk, 0);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_nk = 0u, s_ovf = s_fc[9], s_nout = 0u, s_nq = 0u;
    u32 run = 0u;
    for (int k = 0; k < 8; ++k) {
      s_pre[k] = run;
      run += min(s_fc[k], (u32)FR_RC8);
    }
    s_pre[8] = run;
  }
  __syncthreads();
  const bool dead = s_ovf != 0u;  // an earlier kernel hit a capacity limit: report, leave everything untouched
  const u32 R = s_pre[8];         // tile roots of this search
  if (!dead && R > rcap) return false;  // (uniform; IN_LAUNCH only: FR_RCAP holds every search that is not dead)
  // tile-root id (XCD range | index) <-> dense number
  auto dense_of = [&](u32 gidx) { return s_pre[gidx / FR_RC8] + (gidx & (FR_RC8 - 1u)); };
  auto gid_of = [&](u32 d) {
    u32 xc = 0u;
#pragma unroll
    for (int k = 1; k < 8; ++k) xc += d >= s_pre[k] ? 1u : 0u;
    return xc * FR_RC8 + (d - s_pre[xc]);
  };
  const u32 Rr = (R + RS_T - 1u) / RS_T * RS_T;
  // the pair lists, walked as one sequence; a lane's first three pairs are fetched beside the records
  u32 pp[9];
  pp[0] = 0u;
#pragma unroll
  for (int xc = 0; xc < 8; ++xc) pp[xc + 1] = pp[xc] + min(s_fc[16 + xc], FR_PCAP / 8u);
  auto pair_at = [&](u32 p) {
    u32 xc = 0u;
#pragma unroll
    for (int k = 1; k < 8; ++k) xc += p >= pp[k] ? 1u : 0u;
    return ldw(&F.pairs[(size_t)xc * (FR_PCAP / 8u) + (p - pp[xc])]);
  };
  u32 pk0 = 0u, pk1 = 0u, pk2 = 0u, pk3 = 0u;
  TRec T0, T1;  // records of tile roots d = threadIdx.x and d + RS_T (most searches have fewer than 2 RS_T roots: read once)
  T0.size = T1.size = 0u;
  auto load_rec = [&](u32 gi) -> TRec {
    TRec T;
    if (IN_LAUNCH)
      ld_agent_trec2(&F.trec[gi], &F.trec[gi], T, T);
    else
      T = F.trec[gi];
    return T;
  };
  if (!dead) {
    if (threadIdx.x < pp[8]) pk0 = pair_at(threadIdx.x);
    if (threadIdx.x + RS_T < pp[8]) pk1 = pair_at(threadIdx.x + RS_T);
    if (threadIdx.x + 2 * RS_T < pp[8]) pk2 = pair_at(threadIdx.x + 2 * RS_T);
    if (threadIdx.x + 3 * RS_T < pp[8]) pk3 = pair_at(threadIdx.x + 3 * RS_T);
    u32 nq_part = 0u;
    {  // the first two records of the lane side by side (one round trip)
      const u32 d0 = threadIdx.x, d1 = threadIdx.x + RS_T;
      const u32 g0 = gid_of(min(d0, R ? R - 1u : 0u)), g1 = gid_of(min(d1, R ? R - 1u : 0u));
      const u32 c0 = ldw(&F.tclaim[g0]), c1 = ldw(&F.tclaim[g1]);
      if (IN_LAUNCH)
        ld_agent_trec2(&F.trec[g0], &F.trec[g1], T0, T1);
      else
        T0 = F.trec[g0], T1 = F.trec[g1];
      if (d0 < R) par[d0] = d0, siz[d0] = T0.size, clm[d0] = c0, rko[d0] = T0.own, nq_part += T0.size;
      if (d1 < R) par[d1] = d1, siz[d1] = T1.size, clm[d1] = c1, rko[d1] = T1.own, nq_part += T1.size;
    }
    for (u32 d = threadIdx.x + 2 * RS_T; d < R; d += RS_T) {
      const u32 gi = gid_of(d);
      const TRec T = load_rec(gi);
      par[d] = d;
      siz[d] = T.size;
      clm[d] = ldw(&F.tclaim[gi]);
      rko[d] = T.own;
      nq_part += T.size;
    }
    nq_part = wave_add_u32(nq_part);
    if (lane == 0 && nq_part) atomicAdd(&s_nq, nq_part);
    for (u32 i = threadIdx.x; i < RS_SH; i += RS_T) h_adr[i] = NOCLAIM, h_sum[i] = 0u, h_kept[i] = FR_NOTKEPT;
    for (u32 i = threadIdx.x; i < FR_KCAP * 6; i += RS_T) kbox[i] = (i % 6u) < 3u ? 0xFFFFFFFFu : 0u;
    for (u32 i = threadIdx.x; i < FR_KCAP * 3; i += RS_T) ksum[i] = 0ull;
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 1);
  if (!dead) {
    // ---- cross-tile unions (one list of distinct root pairs per XCD, walked as one sequence so that all
    // lanes are busy at once) ----
    if (threadIdx.x < pp[8]) lds_union_h(par, dense_of(pk0 >> 16), dense_of(pk0 & 0xFFFFu));
    if (threadIdx.x + RS_T < pp[8]) lds_union_h(par, dense_of(pk1 >> 16), dense_of(pk1 & 0xFFFFu));
    if (threadIdx.x + 2 * RS_T < pp[8]) lds_union_h(par, dense_of(pk2 >> 16), dense_of(pk2 & 0xFFFFu));
    if (threadIdx.x + 3 * RS_T < pp[8]) lds_union_h(par, dense_of(pk3 >> 16), dense_of(pk3 & 0xFFFFu));
    for (u32 p = threadIdx.x + 4 * RS_T; p < pp[8]; p += RS_T) {
      const u32 key = pair_at(p);
      lds_union_h(par, dense_of(key >> 16), dense_of(key & 0xFFFFu));
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 2);
  // ---- flatten; cells, lowest claimer and lowest own claimer of every final component (wave-reduced per
  // distinct root) ----
  if (!dead) {
    for (u32 d0 = 0; d0 < Rr; d0 += RS_T) {
      const u32 d = d0 + threadIdx.x;
      const u32 rt = d < R ? lds_find_h(par, d) : 0u;
      const bool mov = d < R && rt != d;
      const u32 sz = mov ? siz[d] : 0u, cl = mov ? clm[d] : NOCLAIM, ow = mov ? rko[d] : NOCLAIM;
      u64 todo = __ballot(mov);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const u32 first = (u32)__shfl((int)rt, leader, 64);
        const bool mine = mov && rt == first;
        u32 s2 = mine ? sz : 0u, c2 = mine ? cl : NOCLAIM, o2 = mine ? ow : NOCLAIM;
        s2 = wave_add_u32(s2), c2 = wave_min_u32(c2), o2 = wave_min_u32(o2);
        if (lane == leader) {
          atomicAdd(&siz[first], s2);
          atomicMin(&clm[first], c2);
          atomicMin(&rko[first], o2);
        }
        todo &= ~__ballot(mine);
      }
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 3);
  if (!dead) {
    // (every find above ran before any sibling was re-parented by another thread's halving in a way that
    // matters: halving only ever installs ancestors; par[] is flat enough for the single look-ups below)
    // ---- clusters: own-claimed components straight to the kept list, seed-claimed ones through the hash.
    // A component is claimed by one of its own cells iff its lowest claimer IS its lowest own claimer (seeds and
    // Q0 cells are disjoint). ----
    for (u32 d = threadIdx.x; d < R; d += RS_T) {
      if (lds_find_h(par, d) != d) continue;
      const u32 cl = clm[d], ow = rko[d];
      if (cl == NOCLAIM) {
        rko[d] = FR_UNCLAIMED;
        continue;
      }
      if (cl == ow) {
        if ((int)siz[d] > F.cluster_min) {
          const u32 e = atomicAdd(&s_nk, 1u);
          if (e < FR_KCAP) {
            k_adr[e] = cl, k_slot[e] = 0u, k_size[e] = siz[d], k_nq[e] = siz[d];
            rko[d] = e;
          } else {
            s_ovf = 1u;
            rko[d] = FR_NOTKEPT;
          }
        } else
          rko[d] = FR_NOTKEPT;
      } else {
        u32 h = (cl * 2654435761u) >> 23;  // 9 bits
        bool done = false;
        for (int probe = 0; probe < RS_SH && !done; ++probe) {
          const u32 old = atomicCAS(&h_adr[h], NOCLAIM, cl);
          if (old == NOCLAIM || old == cl) {
            atomicAdd(&h_sum[h], siz[d]);
            rko[d] = 0x80000000u | h;  // resolved below
            done = true;
          }
          h = (h + 1u) & (RS_SH - 1u);
        }
        if (!done) {
          s_ovf = 1u;
          rko[d] = FR_NOTKEPT;
        }
      }
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 4);
  if (!dead && threadIdx.x < RS_SH && h_adr[threadIdx.x] != NOCLAIM) {
    const u32 cl = h_adr[threadIdx.x], sum = h_sum[threadIdx.x];
    if ((int)(sum + 1u) > F.cluster_min) {
      const u32 e = atomicAdd(&s_nk, 1u);
      if (e < FR_KCAP) {
        k_adr[e] = cl, k_slot[e] = 0xFFFFFFFFu, k_size[e] = sum + 1u, k_nq[e] = sum;
        h_kept[threadIdx.x] = e;
      } else
        s_ovf = 1u;
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 5);
  const u32 nk = min(s_nk, (u32)FR_KCAP);
  const u32 nq = s_nq;
  // (the matrix needs nk * ntx LDS words; beyond that -- hundreds of clusters on a huge map -- the legacy chain)
  const bool bad = dead || s_ovf != 0u || (size_t)nk * (size_t)ntx > 2 * (size_t)rcap || ntx > F.pm_stride;
  if (!bad) {
    // ---- creation order = ascending claimer address; offsets of the grouped cell array ----
    if (threadIdx.x < nk) {
      const u32 ai = k_adr[threadIdx.x];
      u32 rank = 0u, off = 0u;
      for (u32 j = 0; j < nk; ++j)
        if (k_adr[j] < ai) {
          ++rank;
          off += k_nq[j];
        }
      k_rank[threadIdx.x] = rank;
      k_off[threadIdx.x] = off;
      atomicAdd(&s_nout, k_nq[threadIdx.x]);
    }
    // siz / clm are dead from here on: their space becomes the (rank, tile column) matrix
    for (u32 i = threadIdx.x; i < nk * (u32)ntx; i += RS_T) pmx[i] = 0u;
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 6);
  if (!bad) {
    // ---- code of every tile root; index sums / boxes of the kept clusters; cells per (cluster, tile column) ----
    for (u32 d0 = 0; d0 < Rr; d0 += RS_T) {
      const u32 d = d0 + threadIdx.x;
      u32 e = FR_UNCLAIMED, gi = 0u;
      if (d < R) {
        gi = gid_of(d);
        e = rko[lds_find_h(par, d)];
        if (e != FR_UNCLAIMED && e != FR_NOTKEPT && (e & 0x80000000u)) e = h_kept[e & 0x7FFFFFFFu];
      }
      const bool kept = e < FR_KCAP;
      if (d < R) F.rcode[gi] = kept ? k_rank[e] : e;
      CAcc A;
      A.n = 0u, A.sx = A.sy = A.sz = 0u, A.cl = A.lx = A.ly = A.lz = 0xFFFFFFFFu, A.hx = A.hy = A.hz = 0u;
      if (kept) {
        const TRec T = d0 == 0u ? T0 : (d0 == (u32)RS_T ? T1 : load_rec(gi));
        A.sx = T.sx, A.sy = T.sy, A.sz = T.sz, A.lx = T.lo[0], A.ly = T.lo[1], A.lz = T.lo[2];
        A.hx = T.hi[0], A.hy = T.hi[1], A.hz = T.hi[2];
        atomicAdd(&pmx[k_rank[e] * (u32)ntx + T.tx], T.size);
      }
      // the cluster most of the wave's roots belong to (the giant surface) is reduced across the lanes; the few
      // roots of other clusters add themselves (distinct LDS words, no contention to speak of)
      const u64 km = __ballot(kept);
      if (km) {
        u32 first = (u32)__shfl((int)e, __builtin_ctzll(km), 64);
        {  // majority vote between the first two distinct clusters of the wave
          const u64 same = __ballot(kept && e == first);
          const u64 rest = km & ~same;
          if (rest) {
            const u32 second = (u32)__shfl((int)e, __builtin_ctzll(rest), 64);
            if (__popcll(__ballot(kept && e == second)) > __popcll(same)) first = second;
          }
        }
        const bool mine = kept && e == first;
        auto add = [&](u32 ke, const CAcc& B) {
          unsigned long long* qs = ksum + ke * 3u;
          u32* q = kbox + ke * 6u;
          atomicAdd(&qs[0], (unsigned long long)B.sx), atomicAdd(&qs[1], (unsigned long long)B.sy);
          atomicAdd(&qs[2], (unsigned long long)B.sz);
          atomicMin(&q[0], B.lx), atomicMin(&q[1], B.ly), atomicMin(&q[2], B.lz);
          atomicMax(&q[3], B.hx), atomicMax(&q[4], B.hy), atomicMax(&q[5], B.hz);
        };
        if (kept && !mine) add(e, A);
        CAcc B = A;
        if (!mine) B.sx = B.sy = B.sz = 0u, B.lx = B.ly = B.lz = 0xFFFFFFFFu, B.hx = B.hy = B.hz = 0u;
        cacc_reduce(B);
        if (lane == __builtin_ctzll(__ballot(mine))) add(first, B);
      }
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 7);
  if (!bad) {
    // ---- exclusive prefix of every matrix row along the tile columns (a wave per cluster), to memory ----
    for (u32 r = (u32)wave; r < nk; r += RS_T / 64) {
      u32 run = 0u;
      for (int c0 = 0; c0 < ntx; c0 += 64) {
        const int c = c0 + lane;
        const u32 v = c < ntx ? pmx[r * (u32)ntx + (u32)c] : 0u;
        u32 s = v;
        for (int off = 1; off < 64; off <<= 1) {
          const u32 t = (u32)__shfl_up((int)s, off, 64);
          if (lane >= off) s += t;
        }
        if (c < ntx) F.pm[(size_t)r * F.pm_stride + c] = run + s - v;
        run += (u32)__shfl((int)s, 63, 64);
      }
    }
  }
  if (F.dbg && threadIdx.x == 0) {
    F.dbg[(size_t)dblk * FR_DBG_SLOTS + 9] = pp[8];
    F.dbg[(size_t)dblk * FR_DBG_SLOTS + 10] = R;
    F.dbg[(size_t)dblk * FR_DBG_SLOTS + 11] = nq;
  }
  // ---- records: device copy for the kernels that follow, pinned host copy for _search_end ----
  if (!bad && threadIdx.x < nk) {
    const u32 e = threadIdx.x;
    KeptRec r;
    r.addr = k_adr[e], r.slot = k_slot[e], r.size = k_size[e], r.off = k_off[e];
    r.sum[0] = ksum[e * 3u], r.sum[1] = ksum[e * 3u + 1u], r.sum[2] = ksum[e * 3u + 2u];
    for (int k = 0; k < 6; ++k) r.box[k] = kbox[e * 6u + (u32)k];
    r.pad[0] = r.pad[1] = 0u;
    F.krec[k_rank[e]] = r;
    F.h_rec[k_rank[e]] = r;
  }
  if (threadIdx.x == 0) {
    const bool cap = !bad && nq > F.cap_q;  // more Q0 cells than the result arrays hold: an error, not a fallback
    F.counts[0] = min(nq, F.cap_q);
    F.counts[1] = 0u;
    F.counts[2] = bad ? 2u : (cap ? 1u : 0u);  // 2: capacity of the fast path exceeded -> the host runs the legacy chain
    F.counts[6] = dead ? s_fc[9] : (s_ovf ? 16u : (bad ? 17u : 0u));  // which one
    F.counts[3] = bad ? 0u : nk;
    F.counts[5] = bad ? 0u : s_nout;
  }
  if (threadIdx.x < 32) F.fctr[FCTR(threadIdx.x)] = 0u;  // the counters of the NEXT search (its first kernel adds to them at once)
  if (threadIdx.x >= 32 && threadIdx.x < 41) F.fctr[FR_DONE_CTR + FCTR(threadIdx.x - 32)] = 0u;
  if (threadIdx.x == 0) {
    F.counts[8] = V.epoch;              // "resolved" (the kernel k_resolve behind a k_tile_cross that did it returns at once)
    F.counts[9] = IN_LAUNCH ? 1u : 0u;  // ... by whom (fuelmi_frontier_resolved_in_launch)
    F.counts[10] = R;                   // ... and how many tile roots it had (the host's guess for the next search)
  }
  __syncthreads();
  if (threadIdx.x < 15) F.h_counts[threadIdx.x] = F.counts[threadIdx.x];
  // the barrier orders every thread's record stores before thread 0's system-scope release (cumulative): one
  // write-back instead of one per wave
  __syncthreads();
  FR_DBG_MARK(F, dblk, 8);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&F.h_counts[15], V.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // "records are in"
  }
  return true;
}
__global__ void __launch_bounds__(RS_TK) k_resolve(Geo g, FArgs F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ FVar s_v;
  __shared__ u32 s_done;
  {
    constexpr int NV = (int)(sizeof(FVar) / 4);
    const int t = threadIdx.x;
    const u32* src = reinterpret_cast<const u32*>(F.var) + (t < NV ? t : 0);
    if (t == NV) src = F.counts + 8;
    const u32 w = *src;  // (one load instruction for the block and the "resolved" word)
    if (t < NV) reinterpret_cast<u32*>(&s_v)[t] = w;
    if (t == NV) s_done = w;
    __syncthreads();
    if (s_done == s_v.epoch) return;  // the last workgroup of k_tile_cross has resolved this search
  }
  (void)resolve_body<RS_TK, false>(g, F, s_v, smem_raw, FR_RCAP);
}

// Flags (every claimed cell, every NQ seed) and the grouped result: the kept cells cluster by cluster in creation
// order, ascending address inside a cluster.  Address order is (x-row, y, z); a tile column's x-rows interleave the
// column's tiles, so the position of a cell is
//   offset of its cluster + cells of the cluster in the tile columns in front (k_resolve's matrix)
//   + cells of the cluster in the x-rows in front inside this column, over all of its tiles
//   + cells of the cluster in this x-row in the tiles in front (smaller y)        -- both from the per-row counts of
//                                                                                    the column's components
//   + cells of the cluster in front of it in this x-row of this tile (a wave walks the row in order).
// Clusters are told apart inside a tile by a REPRESENTATIVE component (the lowest-numbered component of the tile
// that belongs to the cluster).
template <int NT>
__global__ void __launch_bounds__(NT) k_tile_out(Geo g, FArgs F) {
  FR_DBG_MARK(F, blockIdx.x, 14);  // (before the first load)
  __shared__ TilePro s_pro;
  stage_tile_pro(F, &s_pro);
  const FVar& V = s_pro.V;
  if ((int)blockIdx.x >= V.ntiles_f) return;
  if (s_pro.ovf) return;  // overflow (k_resolve's verdict; not the counter line the atomics went to)
  const TileGeo T = tile_geo(g, V, blockIdx.x);
  const int nseg = T.nseg, items = T.items, TY = T.TY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32* segb = reinterpret_cast<u32*>(smem_raw);  // [items] Q0 bits
  u32* segpre = segb + items;                    // [items + 1]
  u32* segs = segpre + items + 1;                // [items] seed bits
  u32* cadr = segs + items;                      // [FR_TCELL] address of the l-th cell
  u32* rowall = cadr + FR_TCELL;                 // [TX][FR_TROOT] cells of the representative's cluster in x-row lx, whole column
  u32* rowpos = rowall + T.TX * FR_TROOT;        // [TX][FR_TROOT] ... in the tiles in front; then the running output position
  u32* kmap = rowpos + T.TX * FR_TROOT;          // [FR_KCAP] cluster rank -> representative component of this tile (or 0xFFFFFFFF)
  u32* code = kmap + FR_KCAP;                    // [FR_TROOT] rcode of the tile's components
  unsigned short* ckey = reinterpret_cast<unsigned short*>(code + FR_TROOT);  // [FR_TCELL] per cell: representative | 0xFFFE claimed, not kept | 0xFFFF
  __shared__ u32 s_wsum[NT / 64];
  __shared__ u32 s_cn[64], s_cb[64], s_cp[65];  // a chunk of the column's tiles: components, id base, prefix
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dblk = 2 * V.ntiles_f + 2 + (int)blockIdx.x;
  FR_DBG_MARK(F, dblk, 0);
  const u32 nroots = s_pro.nroots, gbase = s_pro.gbase;
  // fetched ahead, beside the bit-planes: the codes of the tile's components, the component numbers of its cells
  // (eight consecutive cells per lane), the first 64 tiles of the column
  u32 my_code = F.rcode[gbase + (threadIdx.x < nroots ? threadIdx.x : 0u)];  // (unconditional, masked)
  if (threadIdx.x >= nroots) my_code = FR_UNCLAIMED;
  const u64 my_tl = reinterpret_cast<const u64*>(F.tlab + (size_t)blockIdx.x * FR_TCELL)[threadIdx.x & (FR_TCELL / 8 - 1)];
  u32 pre_cn, pre_cb;
  {  // (unconditional, masked: see stage_tile_pro)
    const bool ok = threadIdx.x < 64 && (int)threadIdx.x < V.nty_f;
    const int tt = ok ? T.tx * V.nty_f + (int)threadIdx.x : (int)blockIdx.x;
    pre_cn = F.t_nroots[tt], pre_cb = F.t_base[tt];
    if (!ok) pre_cn = 0u, pre_cb = 0u;
  }
  const u32 total = tile_load_arrays<NT, true>(T, F, segb, segpre, s_wsum, segs);  // (uniform)
  FR_DBG_MARK(F, dblk, 1);
  // ---- NQ seeds are flagged whatever happens to the components around them ----
  auto or_flags = [&](long a0, u32 bits) {  // bits of the 32 voxels from address a0
    if (!bits) return;
    const long w = a0 >> 6;
    const int sh = (int)(a0 & 63);
    atomicOr(reinterpret_cast<unsigned long long*>(&F.flag[w]), (unsigned long long)bits << sh);
    if (sh > 32) atomicOr(reinterpret_cast<unsigned long long*>(&F.flag[w + 1]), (unsigned long long)bits >> (64 - sh));
  };
  if (total == 0u || nroots == 0u) {
    for (int it = threadIdx.x; it < items; it += NT) {
      const u32 sb = segs[it];
      if (!sb) continue;
      const int line = it / nseg, c = it - line * nseg;
      or_flags(tile_line_adr(g, T, line) + 32 * c, sb);
    }
    return;
  }
  // ---- codes of the tile's components, representatives ----
  for (int k = threadIdx.x; k < FR_KCAP; k += NT) kmap[k] = 0xFFFFFFFFu;
  for (int k = threadIdx.x; k < T.TX * FR_TROOT; k += NT) rowall[k] = 0u, rowpos[k] = 0u;
  if (threadIdx.x < nroots) code[threadIdx.x] = my_code;
  __syncthreads();
  if (threadIdx.x < nroots && code[threadIdx.x] < FR_KCAP) atomicMin(&kmap[code[threadIdx.x]], threadIdx.x);
  // ---- cells: address, key ----
  if (threadIdx.x < FR_TCELL / 8) {  // (component number for now)
#pragma unroll
    for (int q = 0; q < 8; ++q) ckey[8 * threadIdx.x + q] = (unsigned short)((my_tl >> (8 * q)) & 0xFFull);
  }
  for (int it = threadIdx.x; it < items; it += NT) {
    u32 rem = segb[it];
    if (!rem) continue;
    const int line = it / nseg, c = it - line * nseg;
    const u32 a0 = (u32)(tile_line_adr(g, T, line) + 32 * c);
    u32 l = segpre[it];
    while (rem) {
      const int b = __builtin_ctz(rem);
      rem &= rem - 1u;
      cadr[l++] = a0 + (u32)b;
    }
  }
  __syncthreads();
  for (u32 l = threadIdx.x; l < total; l += NT) {
    const u32 cd = code[ckey[l]];
    ckey[l] = (unsigned short)(cd < FR_KCAP ? kmap[cd] : (cd == FR_UNCLAIMED ? 0xFFFFu : 0xFFFEu));
  }
  FR_DBG_MARK(F, dblk, 2);
  // ---- the column: per-row counts of every component of every tile of column tx, folded into the
  // representatives of this tile's clusters ----
  const int nty = V.nty_f;
  for (int t0 = 0; t0 < nty; t0 += 64) {
    const int nt = min(64, nty - t0);
    __syncthreads();
    if (threadIdx.x < 64) {
      u32 cn = pre_cn, cb = pre_cb;
      if (t0 > 0) {
        cn = cb = 0u;
        if ((int)threadIdx.x < nt) {
          const int tt = T.tx * nty + t0 + threadIdx.x;
          cn = F.t_nroots[tt], cb = F.t_base[tt];
        }
      }
      u32 s = cn;
      for (int off = 1; off < 64; off <<= 1) {
        const u32 t = (u32)__shfl_up((int)s, off, 64);
        if (lane >= off) s += t;
      }
      s_cn[threadIdx.x] = cn, s_cb[threadIdx.x] = cb, s_cp[threadIdx.x] = s - cn;
      if (threadIdx.x == 63) s_cp[64] = s;
    }
    __syncthreads();
    const u32 ncomp = s_cp[64];
    for (u32 j = threadIdx.x; j < ncomp; j += NT) {
      int lo = 0, hi = nt - 1;  // tile of the j-th component of the chunk
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_cp[mid] <= j)
          lo = mid;
        else
          hi = mid - 1;
      }
      const u32 gi = s_cb[lo] + (j - s_cp[lo]);
      const u32 cd = F.rcode[gi];
      if (cd >= FR_KCAP) continue;
      const u32 rep = kmap[cd];
      if (rep == 0xFFFFFFFFu) continue;  // a cluster this tile has no cell of
      const uint4* rw = reinterpret_cast<const uint4*>(F.rrow + (size_t)gi * FR_TXS);
      const uint4 r0 = rw[0], r1 = rw[1];
      const u32 pk[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      const bool front = t0 + lo < T.ty;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const u32 c0 = pk[q] & 0xFFFFu, c1 = pk[q] >> 16;
        if (c0 && 2 * q < T.TX) {
          atomicAdd(&rowall[(2 * q) * FR_TROOT + rep], c0);
          if (front) atomicAdd(&rowpos[(2 * q) * FR_TROOT + rep], c0);
        }
        if (c1 && 2 * q + 1 < T.TX) {
          atomicAdd(&rowall[(2 * q + 1) * FR_TROOT + rep], c1);
          if (front) atomicAdd(&rowpos[(2 * q + 1) * FR_TROOT + rep], c1);
        }
      }
    }
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 3);
  // ---- first output position of every (x-row, representative) ----
  for (u32 j = threadIdx.x; j < nroots * (u32)T.TX; j += NT) {
    const u32 rep = j / (u32)T.TX, lx = j - rep * (u32)T.TX;
    const u32 cd = code[rep];
    if (cd >= FR_KCAP || kmap[cd] != rep) continue;
    u32 pos = F.krec[cd].off + F.pm[(size_t)cd * F.pm_stride + T.tx];
    for (u32 q = 0; q < lx; ++q) pos += rowall[q * FR_TROOT + rep];
    rowpos[lx * FR_TROOT + rep] += pos;
  }
  __syncthreads();
  FR_DBG_MARK(F, dblk, 4);
  // ---- the rows: a wave walks the cells of an x-row in address order, 64 at a time ----
  u32* key_out = F.ms_key[1];
  u32* val_out = F.ms_val[1];
  const bool direct = F.counts[0] <= F.hcells_direct_max;  // big lists are fetched on demand (0.5 MB over PCIe
                                                           // per search otherwise, whether anybody reads it or not)
  for (int lx = wave; lx < T.nxl; lx += NT / 64) {
    const u32 l0 = segpre[lx * TY * nseg], l1 = segpre[(lx + 1) * TY * nseg];
    for (u32 lb = l0; lb < l1; lb += 64u) {
      const u32 l = lb + (u32)lane;
      const u32 kk = l < l1 ? (u32)ckey[l] : 0xFFFFu;
      const bool active = kk < FR_TROOT;
      u32 pos = 0u;
      u64 todo = __ballot(active);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const u32 dl = (u32)__shfl((int)kk, leader, 64);
        const u64 same = __ballot(active && kk == dl);
        volatile u32* rp = &rowpos[(u32)lx * FR_TROOT + dl];  // (other lanes' stores must be seen: no caching in registers)
        const u32 base = *rp;
        if (active && kk == dl) pos = base + (u32)__popcll(same & ((1ull << lane) - 1ull));
        if (lane == leader) *rp = base + (u32)__popcll(same);
        todo &= ~same;
      }
      if (active) {
        const u32 a = cadr[l];
        key_out[pos] = code[kk];
        val_out[pos] = a;
        if (direct) F.h_cells[pos] = a;  // posted write over PCIe
      }
    }
  }
  FR_DBG_MARK(F, dblk, 5);
  // ---- flags: claimed cells + seeds, one or two atomics per 32-voxel segment ----
  for (int it = threadIdx.x; it < items; it += NT) {
    u32 rem = segb[it];
    u32 fl = segs[it];
    u32 l = segpre[it];
    while (rem) {
      const int b = __builtin_ctz(rem);
      rem &= rem - 1u;
      if (ckey[l++] != 0xFFFFu) fl |= 1u << b;
    }
    if (!fl) continue;
    const int line = it / nseg, c = it - line * nseg;
    or_flags(tile_line_adr(g, T, line) + 32 * c, fl);
  }  FR_DBG_MARK(F, dblk, 6);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static const int kFastMenu[4][2] = {{16, 32}, {8, 32}, {8, 16}, {4, 8}};  // tiles of the fast chain (x-rows, z-lines)
static inline int fblocks(long n, int t, int cap = 1 << 16) {
  long b = (n + t - 1) / t;
  return (int)std::max(1L, std::min((long)cap, b));
}


static bool have_overlap(const double* min1, const double* max1, const double* min2, const double* max2) {
  // haveOverlap (:353-363)
  for (int i = 0; i < 3; ++i) {
    double bmin = std::max(min1[i], min2[i]);
    double bmax = std::min(max1[i], max2[i]);
    if (bmin > bmax + 1e-3) return false;
  }
  return true;
}

template <typename T>
static int dmalloc(fuelmi_frontier* f, T** p, size_t n) {
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)));
  f->allocs.push_back(d);
  *p = (T*)d;
  const char* fp = reinterpret_cast<const char*>(p), *f0 = reinterpret_cast<const char*>(&f->F);
  if (fp >= f0 && fp < f0 + sizeof(FArgs))  // a per-search buffer of F: the other plane's set gets a twin (frontier_twin_set)
    f->f_scratch.push_back({(size_t)(fp - f0), std::max<size_t>(n, 1) * sizeof(T)});
  return FUELMI_OK;
}
// priority of the finder's streams: the highest (a chain of short latency-bound kernels beside the wide ESDF passes;
// normal / lowest measured in rounds 3, 4 and 6: within the run-to-run spread or worse, profiles/r06_frontier_prio_threads_sweep2.txt)
static int frontier_stream_priority() {
  int lo_p = 0, hi_p = 0;
  if (hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) != hipSuccess) return 0;
  return hi_p;
}
// the second set of per-search buffers (see fuelmi_frontier::F2): a twin of every device buffer F points to, its own
// per-search variables and its own pinned result block
static int frontier_twin_set(fuelmi_frontier* f) {
  FArgs& F = f->F;
  f->F2 = F;
  char* f2 = reinterpret_cast<char*>(&f->F2);
  const std::vector<fuelmi_frontier::ScratchRec> recs = f->f_scratch;  // (dmalloc below appends)
  for (const auto& r : recs) {
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, r.bytes));
    f->allocs.push_back(d);
    memcpy(f2 + r.field_off, &d, sizeof(void*));
  }
  f->f_scratch = recs;
  FArgs& G2 = f->F2;
  G2.kept = G2.counts + 16;
  HIPCHK(hipMemsetAsync(G2.counts, 0, 16 * sizeof(u32), f->stream));
  HIPCHK(hipMemsetAsync(G2.fctr, 0, FR_NCTR * sizeof(u32), f->stream));
  FVar* d_var2 = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&d_var2), sizeof(FVar)));
  f->allocs.push_back(d_var2);
  G2.var = d_var2, G2.var_w = d_var2;
  HIPCHK(hipHostMalloc(&f->h_pin2, f->pin_bytes, hipHostMallocDefault));
  G2.h_counts = reinterpret_cast<u32*>(f->h_pin2);
  G2.h_rec = reinterpret_cast<KeptRec*>(G2.h_counts + 16);
  G2.h_part = reinterpret_cast<u32*>(G2.h_rec + G2.cap_kept);
  G2.h_cells = G2.h_part + ((size_t)G2.cap_q / SZ_CH + 2) * 10;
  memset(f->h_pin2, 0, 64);
  G2.flag = f->flag2.p;
  HIPCHK(fuelmi_stream_create(&f->stream2, frontier_stream_priority(), "FR"));
  HIPCHK(hipStreamSynchronize(f->stream));
  return FUELMI_OK;
}

static int frontier_ensure_stage(fuelmi_frontier* f, size_t bytes) {
  if (bytes > f->d_stage_bytes) {
    if (f->d_stage) HIPCHK(hipFree(f->d_stage));
    f->d_stage = nullptr;
    f->d_stage_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc(&f->d_stage, want));
    f->d_stage_bytes = want;
  }
  return FUELMI_OK;
}

// both streams of the finder (the current search's and the one the previous fresh search left its tail on)
static hipError_t frontier_drain(const fuelmi_frontier* f) {
  const hipError_t e = stream_wait(f->stream);
  if (e != hipSuccess || !f->stream2) return e;
  return stream_wait(f->stream2);
}
static void frontier_orphan(void* p) {  // the map is going away under a live finder
  fuelmi_frontier* f = static_cast<fuelmi_frontier*>(p);
  if (f->stream) (void)hipStreamSynchronize(f->stream);
  if (f->stream2) (void)hipStreamSynchronize(f->stream2);
  f->scope.reset();
  f->map = nullptr;
}
extern "C" void fuelmi_frontier_destroy(fuelmi_frontier* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->map) {
    (void)hipStreamSynchronize(f->map->stream);
    std::lock_guard<std::mutex> lk(f->map->dep_mu);
    auto& deps = f->map->dependents;
    for (size_t k = 0; k < deps.size(); ++k)
      if (deps[k].obj == f) {
        deps.erase(deps.begin() + (long)k);
        break;
      }
  }
  for (hipStream_t st : {f->stream, f->stream2})
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  if (f->h_pin2) (void)hipHostFree(f->h_pin2);
  if (f->ev_dep) (void)hipEventDestroy(f->ev_dep);
  if (f->d_stage) (void)hipFree(f->d_stage);
  for (void* p : f->allocs) (void)hipFree(p);
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  if (f->h_var) (void)hipHostFree(f->h_var);
  if (f->pool) (void)hipFree(f->pool);
  if (f->h_changed) (void)hipHostFree(f->h_changed);
  if (f->h_cand) (void)hipHostFree(f->h_cand);
  if (f->h_put) (void)hipHostFree(f->h_put);
  if (f->d_mark) (void)hipFree(f->d_mark);
  if (f->rm_bar) (void)hipFree(f->rm_bar);
  frontier_split_free(f);
  frontier_order_free(f);
  for (auto& row : f->graph_exec)
    for (hipGraphExec_t e : row)
      if (e) (void)hipGraphExecDestroy(e);
  if (f->ev_tail) (void)hipEventDestroy(f->ev_tail);
  if (f->ev_prev) (void)hipEventDestroy(f->ev_prev);
  if (f->ev_planes_read) {
    if (f->map) map_drop_plane_reader(f->map, f->ev_planes_read), map_drop_late_reader(f->map, f->ev_planes_read);
    (void)hipEventDestroy(f->ev_planes_read);
  }
  Plane* pl[] = {&f->flag, &f->flag2, &f->qb, &f->sb};
  for (Plane* p : pl)
    if (p->base) (void)hipFree(p->base);
  delete f;
}

extern "C" int fuelmi_frontier_create(fuelmi_map* m, const fuelmi_frontier_cfg* cfg, fuelmi_frontier** out) {
  ARGCHK(m && cfg && out);
  ARGCHK(cfg->reference_order >= 0 && cfg->reference_order <= 2);
  *out = nullptr;
  HIPCHK(hipSetDevice(m->device));
  fuelmi_frontier* f = new fuelmi_frontier;
  f->map = m;
  f->device = m->device;
  f->cfg = *cfg;
  const Geo& g = m->g;
  // first z index whose centre is NOT below min_z (reference: if (pos[2] < 0.4) continue;)
  f->iz_min = g.nz;
  for (int iz = 0; iz < g.nz; ++iz) {
    double pz = (iz + 0.5) * g.res + g.org[2];
    if (!(pz < cfg->min_z)) {
      f->iz_min = iz;
      break;
    }
  }
  int rc;
  if ((rc = plane_alloc(m, f->flag)) || (rc = plane_alloc(m, f->flag2)) || (rc = plane_alloc(m, f->qb)) ||
      (rc = plane_alloc(m, f->sb))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  FArgs& F = f->F;
  memset(&F, 0, sizeof(F));
  size_t nwords = ((size_t)g.W + 255) / 256 * 256 + 256;
  F.cap_q = (u32)std::max<size_t>(1u << 20, (size_t)g.N / 8);
  F.cap_s = (u32)std::max<size_t>(1u << 18, (size_t)g.N / 32);
  F.cap_kept = 1u << 16;
  F.cluster_min = cfg->cluster_min;
  if ((rc = dmalloc(f, &F.pref, nwords)) || (rc = dmalloc(f, &F.blocksum, nwords / 256 + 1)) ||
      (rc = dmalloc(f, &F.blockscan, nwords / 256 + 1)) ||
      (rc = dmalloc(f, &F.counts, 16 + (size_t)F.cap_kept * 3)) ||  // counts, then the kept list
      (rc = dmalloc(f, &F.cell_adr, F.cap_q)) || (rc = dmalloc(f, &F.parent, F.cap_q)) ||
      (rc = dmalloc(f, &F.claim, F.cap_q)) || (rc = dmalloc(f, &F.cell_slot, F.cap_q)) ||
      (rc = dmalloc(f, &F.seed_adr, F.cap_s)) || (rc = dmalloc(f, &F.csize, (size_t)F.cap_q + F.cap_s)) ||
      (rc = dmalloc(f, &F.slot2rank, (size_t)F.cap_q + F.cap_s)) || (rc = dmalloc(f, &F.krec, F.cap_kept)) ||
      (rc = dmalloc(f, &F.ms_key[0], F.cap_q)) || (rc = dmalloc(f, &F.ms_key[1], F.cap_q)) ||
      (rc = dmalloc(f, &F.ms_val[0], F.cap_q)) || (rc = dmalloc(f, &F.ms_val[1], F.cap_q)) ||
      (rc = dmalloc(f, &F.ms_hist, (size_t)256 * (F.cap_q / MS_CH + 2))) ||
      (rc = dmalloc(f, &F.info_part, ((size_t)F.cap_q / SZ_CH + 2) * 10))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  F.kept = F.counts + 16;
  HIPCHK(hipMemsetAsync(F.counts, 0, 16 * sizeof(u32), m->stream));  // ([8]: epoch of the last search resolved; no search has epoch 0)
  F.keys_from_slots = 1;
  if ((rc = dmalloc(f, &F.trec, FR_RCAP + 256)) || (rc = dmalloc(f, &F.tclaim, FR_RCAP + 256)) || (rc = dmalloc(f, &F.rcode, FR_RCAP + 256)) ||
      (rc = dmalloc(f, &F.rrow, (size_t)FR_RCAP * FR_TXS)) || (rc = dmalloc(f, &F.pairs, (size_t)FR_PCAP)) ||
      (rc = dmalloc(f, &F.fctr, FR_NCTR))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  HIPCHK(hipMemsetAsync(F.fctr, 0, FR_NCTR * sizeof(u32), m->stream));
  F.dbg = nullptr;
  if (getenv("FUELMI_FR_TIMING")) {  // dev aid: phase time stamps of the fast chain (one row per tile + one for k_resolve)
    const size_t nrow = (size_t)(3 * ((g.nx + 4) / 4) * ((g.ny + 8) / 8) + 64);
    if ((rc = dmalloc(f, &F.dbg, nrow * FR_DBG_SLOTS))) {
      fuelmi_frontier_destroy(f);
      return rc;
    }
    HIPCHK(hipMemsetAsync(F.dbg, 0, nrow * FR_DBG_SLOTS * sizeof(unsigned long long), m->stream));
  }
  F.occ = m->occ_bits.p;
  F.unk = m->unk_bits.p;
  F.flag = f->flag.p;
  F.qb = f->qb.p;
  F.sb = f->sb.p;
  HIPCHK(hipStreamSynchronize(m->stream));
  {
    // the scan is a chain of short latency-bound kernels and is the critical path of a plan cycle:
    // give its stream the highest priority so the wide ESDF kernels of the map stream fill in around it
    HIPCHK(fuelmi_stream_create(&f->stream, frontier_stream_priority(), "FR"));
  }
  HIPCHK(hipEventCreateWithFlags(&f->ev_dep, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&f->ev_tail, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&f->ev_prev, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&f->ev_planes_read, hipEventDisableTiming));

  // ---- everything below is constant for the life of the object (the kernel chain is replayed
  // as a graph with these arguments baked in) ----
  // Q box: isInBox(idx) (min <= id < max) and z >= iz_min, inside the map
  const int nv[3] = {g.nx, g.ny, g.nz};
  for (int k = 0; k < 3; ++k) {
    F.qbox.lo[k] = std::max(m->info.box_min[k], 0);
    F.qbox.hi[k] = std::min(m->info.box_max[k] - 1, nv[k] - 1);
  }
  F.qbox.lo[2] = std::max(F.qbox.lo[2], f->iz_min);
  for (int k = 0; k < 3; ++k)
    if (F.qbox.lo[k] > F.qbox.hi[k]) {  // degenerate exploration box: nothing can be added
      F.qbox.lo[k] = 1;
      F.qbox.hi[k] = 0;
    }
  // per-search arguments: pinned host copy + device copy
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_var), sizeof(FVar), hipHostMallocDefault));
  memset(f->h_var, 0, sizeof(FVar));
  if ((rc = dmalloc(f, &f->d_var, 1))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  F.var = f->d_var;
  F.var_w = f->d_var;
  // results land in one pinned host buffer [counts | cluster records | chunk records | cells]
  f->pin_bytes = 64 + (size_t)F.cap_kept * sizeof(KeptRec) + ((size_t)F.cap_q / SZ_CH + 2) * 40 + (size_t)F.cap_q * 4;
  HIPCHK(hipHostMalloc(&f->h_pin, f->pin_bytes, hipHostMallocDefault));
  F.h_counts = reinterpret_cast<u32*>(f->h_pin);
  F.h_rec = reinterpret_cast<KeptRec*>(F.h_counts + 16);
  F.h_part = reinterpret_cast<u32*>(F.h_rec + F.cap_kept);
  F.h_cells = F.h_part + ((size_t)F.cap_q / SZ_CH + 2) * 10;
  F.hcells_direct_max = FR_HCELLS_DIRECT;
  // CCL tile = TX x TY z-lines with u32 labels in LDS (<= 48 KiB so three workgroups share a CU)
  f->TY = 16;
  f->TX = std::max(1, std::min(8, (48 * 1024) / (f->TY * g.nz * 4)));
  const int qx = F.qbox.hi[0] - F.qbox.lo[0] + 1, qy = F.qbox.hi[1] - F.qbox.lo[1] + 1;
  f->ccl_tiles = 0;
  if (qx > 0 && qy > 0 && F.qbox.lo[2] <= F.qbox.hi[2]) {
    const int TX = f->TX, TY = f->TY;
    const int ntx = (qx + TX - 1) / TX, nty = (qy + TY - 1) / TY;
    f->ccl_tiles = ntx * nty;
    f->ccl_lds = ((size_t)TX * TY * g.nz + 2 * (size_t)TX * TY * ((g.nz + 31) / 32) + 2 * (size_t)TX) * sizeof(u32);
    if (f->ccl_lds > 160 * 1024) {
      fuelmi_set_error("frontier CCL tile of %d z-lines x %d voxels does not fit the LDS", TX * TY, g.nz);
      fuelmi_frontier_destroy(f);
      return FUELMI_ELIMIT;
    }
    if (f->ccl_lds > 64 * 1024)
    {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_local<512>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->ccl_lds));
    }
    // fast path: sparse labels -- the tile is not bound by nz.  32 z-lines deep, 8 wide, 16 wide once that
    // still leaves >= 512 tiles (measured on 400^2 x 100 and 800^2 x 200 maps: fewer tile roots and face pairs
    // for k_tile_cross / k_resolve outweigh the longer tiles).
    // The tile of a search is picked from a menu by the size of its region; the launch grid of a menu entry is
    // that tile over the largest rectangle a search can cover (the Q box plus the box_max face of the scan box).
    const int mtx = 4, mty = 8;  // smallest tile of the menu
    f->fast_tiles = ((qx + 1 + mtx - 1) / mtx) * ((qy + 1 + mty - 1) / mty);
    F.pm_stride = (qx + 1 + mtx - 1) / mtx;
    if ((rc = dmalloc(f, &F.vlab, (size_t)g.N + 64)) || (rc = dmalloc(f, &F.tlab, (size_t)f->fast_tiles * FR_TCELL)) ||
        (rc = dmalloc(f, &F.t_base, (size_t)f->fast_tiles)) || (rc = dmalloc(f, &F.t_nroots, (size_t)f->fast_tiles)) ||
        (rc = dmalloc(f, &F.pm, (size_t)FR_KCAP * F.pm_stride))) {
      fuelmi_frontier_destroy(f);
      return rc;
    }
    size_t seg_words = 0;  // per-tile segment arrays: the largest (tiles x segments per tile) of the menu
    for (int k = 0; k < 4; ++k) {
      const int ftx = kFastMenu[k][0], fty = kFastMenu[k][1];
      const size_t items = (size_t)(ftx * fty) * ((g.nz + 31) / 32);
      seg_words = std::max(seg_words, (size_t)((qx + 1 + ftx - 1) / ftx) * ((qy + 1 + fty - 1) / fty) * items);
      f->fast_items[k] = items;
      // k_tile_ccl: labels, bits + prefix, records, per-row counts, root numbers, component per cell
      f->tile_lds[k] = (FR_TCELL + 3 * items + 1 + (size_t)FR_TROOT * 8 + (size_t)FR_TROOT * FR_TXS) * sizeof(u32) +
                       FR_TCELL * sizeof(unsigned short) + FR_TCELL;
      f->cross_lds[k] = (2 * items + (size_t)XC_WCAP) * sizeof(u32);
      // k_tile_out: bits + prefix + seed bits, cell addresses, the two row tables, cluster map, codes, keys
      f->out_lds[k] = (3 * items + 1 + FR_TCELL + 2 * (size_t)kFastMenu[k][0] * FR_TROOT + FR_KCAP + FR_TROOT) * sizeof(u32) +
                      FR_TCELL * sizeof(unsigned short);
      f->tile_lds[k] = (f->tile_lds[k] + 15) & ~(size_t)15;
      f->cross_lds[k] = (f->cross_lds[k] + 15) & ~(size_t)15;
      f->out_lds[k] = (f->out_lds[k] + 15) & ~(size_t)15;
    }
    if ((rc = dmalloc(f, &F.tq, seg_words + 64)) || (rc = dmalloc(f, &F.ts, seg_words + 64))) {
      fuelmi_frontier_destroy(f);
      return rc;
    }
    const size_t lds_max = std::max(f->tile_lds[0], f->out_lds[0]);
    if (lds_max > 64 * 1024 && lds_max <= 150 * 1024) {
      // (the attribute belongs to the FUNCTION, not to this finder: always the kernels' ceiling, so that a second
      // finder with a shorter map cannot lower it under a taller finder's launches -- ADVICE r3)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_ccl<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_out<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
  }
  f->resolve_lds = resolve_lds_bytes(FR_RCAP);
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)f->resolve_lds));
  // the fast path needs tiles, a cluster threshold that rules out one-seed clusters, and tile-local indices
  // that fit the 16-bit root numbers
  // (a thread of the tile kernels fetches at most FT_PER segments: z-lines of up to 256 voxels with the largest tile)
  f->fast_ok = f->ccl_tiles > 0 && cfg->cluster_min >= 1 && std::max(f->tile_lds[0], f->out_lds[0]) <= 150 * 1024 &&
               f->fast_items[0] <= (size_t)FT_PER * 512 && getenv("FUELMI_FRONTIER_LEGACY") == nullptr;
  if ((rc = frontier_twin_set(f))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(m->dep_mu);
    m->dependents.push_back({f, &frontier_orphan});
  }
  *out = f;
  return FUELMI_OK;
}

// copy the cell lists of kept-but-still-lazy clusters out of the pinned result buffer (see frontier_keep_clusters)
int frontier_materialize_lists(fuelmi_frontier* f) {
  // (the name is round 2's: the lists are no longer copied out here -- see HCluster::dev_only)
  if (!f->lazy_kept) return FUELMI_OK;
  for (std::list<HCluster>* L : {&f->frontiers, &f->dormant})
    for (HCluster& c : *L)
      if (c.lazy) {
        c.dev_n = (u32)c.size();
        c.lazy = nullptr;
        c.lazy_n = 0;
        c.dev_only = true;
      }
  f->lazy_kept = false;
  return FUELMI_OK;
}
// the host list of a cluster that lives in the device pool only: fetched when somebody asks for it
static int frontier_fetch_cluster(const fuelmi_frontier* f, const HCluster* cc) {
  HCluster* c = const_cast<HCluster*>(cc);
  if (!c->dev_only) return FUELMI_OK;
  HIPCHK(hipStreamSynchronize(f->stream));  // (the copy into the pool is queued on the finder's stream)
  std::vector<int> v(c->dev_n);
  if (c->dev_n) HIPCHK(hipMemcpy(v.data(), f->pool + c->pool_off, (size_t)c->dev_n * sizeof(int), hipMemcpyDeviceToHost));
  if (c->lazy_seed >= 0 && !v.empty()) {  // the NQ seed travels behind the sorted cells: back into address order
    const int seed = v.back();
    v.pop_back();
    v.insert(std::lower_bound(v.begin(), v.end(), seed), seed);
  }
  c->cells.swap(v);
  c->lazy_seed = -1;
  c->dev_only = false;
  return FUELMI_OK;
}

// ---- device pool of committed clusters' cells ---------------------------------------------------
static int pool_upload(fuelmi_frontier* f, HCluster& c) {  // from the host list (rebuilds)
  c.pool_off = f->pool_used;
  if (!c.cells.empty())
    HIPCHK(hipMemcpyAsync(f->pool + f->pool_used, c.cells.data(), c.cells.size() * sizeof(int), hipMemcpyHostToDevice,
                          f->stream));
  f->pool_used += c.cells.size();
  return FUELMI_OK;
}
static int pool_reserve(fuelmi_frontier* f, size_t need) {
  if (f->pool_used + need <= f->pool_cap) return FUELMI_OK;
  // compact (erased clusters leave holes) and grow: re-upload the live clusters from their host lists
  size_t live = need;
  {
    const int rcm = frontier_materialize_lists(f);
    if (rcm) return rcm;
  }
  for (std::list<HCluster>* L : {&f->frontiers, &f->dormant})
    for (HCluster& c : *L) {
      const int rcf = frontier_fetch_cluster(f, &c);  // (the pool is about to be rebuilt from the host lists)
      if (rcf) return rcf;
      live += c.cells.size();
    }
  HIPCHK(hipStreamSynchronize(f->stream));
  if (live > f->pool_cap / 2 || !f->pool) {
    size_t cap = std::max<size_t>(1u << 20, f->pool_cap);
    while (cap / 2 < live) cap *= 2;
    if (f->pool) HIPCHK(hipFree(f->pool));
    f->pool = nullptr;
    f->pool_cap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&f->pool), cap * sizeof(u32)));
    f->pool_cap = cap;
  }
  f->pool_used = 0;
  for (std::list<HCluster>* L : {&f->frontiers, &f->dormant})
    for (HCluster& c : *L) {
      int rc = pool_upload(f, c);
      if (rc) return rc;
    }
  HIPCHK(hipStreamSynchronize(f->stream));  // the host lists may be freed by the caller afterwards
  return FUELMI_OK;
}
// commit this search's clusters to the pool: the ones whose cells still sit grouped on the device are
// copied there by ONE launch (a table of {destination, source, count, seed} per cluster)
int frontier_keep_clusters(fuelmi_frontier* f, std::list<HCluster>& clusters) {
  // The kept clusters stay "lazy" -- their cell lists still sit in the pinned result buffer, which the tail of
  // the search may not even have filled yet: the pool copy below is ordered behind that tail on the stream, and
  // the host lists are materialised by frontier_materialize_lists before the buffer is reused (next search) or
  // when somebody asks for them.  Waiting for the tail here cost ~25 us per streaming cycle.
  size_t need = 0, nlazy = 0;
  for (HCluster& c : clusters) need += c.size(), nlazy += c.lazy ? 1 : 0;
  f->pool_dirty = true;
  int rc = pool_reserve(f, need);
  if (rc) return rc;
  std::vector<PoolPut> table;
  table.reserve(nlazy);
  for (HCluster& c : clusters) {
    if (!c.lazy) {
      if ((rc = pool_upload(f, c))) return rc;
      continue;
    }
    // one workgroup per table entry: a large cluster (the growing surface of a streaming run reaches 20 k cells) is
    // cut into pieces of 2048 cells -- one workgroup walking it alone was 13 us of every frame's frontier stream
    const u32 src0 = (u32)(c.lazy - reinterpret_cast<const int*>(f->F.h_cells)), ncell = (u32)c.lazy_n;
    for (u32 at = 0; at < ncell || at == 0; at += 2048u) {
      PoolPut e;
      e.dst = f->pool_used + at;
      e.src = src0 + at;
      e.n = std::min(2048u, ncell - at);
      e.seed = at + 2048u >= ncell ? c.lazy_seed : -1;  // (the seed goes behind the last piece)
      e.pad = 0;
      table.push_back(e);
      if (ncell == 0) break;
    }
    c.pool_off = f->pool_used;
    f->pool_used += c.size();
    f->lazy_kept = true;
  }
  if (table.empty()) return FUELMI_OK;
  if (table.size() > f->h_put_cap) {
    if (f->h_put) HIPCHK(hipHostFree(f->h_put));
    f->h_put = nullptr;
    f->h_put_cap = 0;
    const size_t cap = table.size() + table.size() / 2 + 64;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_put), cap * sizeof(PoolPut), hipHostMallocDefault));
    f->h_put_cap = cap;
  }
  // (the table of the previous commit was consumed before the search in between was collected)
  memcpy(f->h_put, table.data(), table.size() * sizeof(PoolPut));
  k_pool_put<<<(unsigned)table.size(), 256, 0, f->stream>>>(f->pool, f->F.ms_val[f->last_fin],
                                                             reinterpret_cast<const PoolPut*>(f->h_put));
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

// Drop the clusters that overlap the updated box and contain a cell that is no longer a frontier cell
// (searchFrontiers :62-93).  The test and the clearing of the flags run on the device ahead of the
// scan; the host learns the verdicts together with the search result (no round trip of its own) and
// updates frontiers_ / dormant_frontiers_ / removed_ids_ in _search_end.  The search region therefore
// includes the boxes of ALL candidates, not only of the ones that turn out to be dropped.
static int remove_changed_begin(fuelmi_frontier* f, const double* umin, const double* umax) {
  fuelmi_map* m = f->map;
  f->pend_rm.clear();
  for (std::list<HCluster>* L : {&f->frontiers, &f->dormant}) {
    int pos = 0;
    for (auto it = L->begin(); it != L->end(); ++it, ++pos)
      if (have_overlap(it->bmin, it->bmax, umin, umax)) f->pend_rm.push_back({L, it, pos});
  }
  const size_t nc = f->pend_rm.size();
  if (nc == 0) return FUELMI_OK;
  std::vector<u64> off(nc);
  std::vector<u32> start(nc);
  u32 total = 0;
  for (size_t k = 0; k < nc; ++k) {
    const HCluster& c = *f->pend_rm[k].it;
    off[k] = c.pool_off;
    start[k] = total;
    total += (u32)c.size();
    for (int q = 0; q < 3; ++q) {  // if dropped, its cells lose their flags and may be re-grown from the scan box
      const int lo = (int)std::floor((c.bmin[q] - m->g.org[q]) * m->g.res_inv);
      const int hi = (int)std::floor((c.bmax[q] - m->g.org[q]) * m->g.res_inv);
      if (f->rm_lo[q] > f->rm_hi[q])
        f->rm_lo[q] = lo, f->rm_hi[q] = hi;
      else
        f->rm_lo[q] = std::min(f->rm_lo[q], lo), f->rm_hi[q] = std::max(f->rm_hi[q], hi);
    }
  }
  if (nc > f->h_changed_cap) {
    if (f->h_changed) HIPCHK(hipHostFree(f->h_changed));
    if (f->h_cand) HIPCHK(hipHostFree(f->h_cand));
    if (f->d_mark) HIPCHK(hipFree(f->d_mark));
    f->h_changed = nullptr;
    f->h_cand = nullptr;
    f->d_mark = nullptr;
    f->h_changed_cap = 0;
    const size_t cap = nc + nc / 2 + 64;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_changed), cap * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_cand), cap * sizeof(RmCand), hipHostMallocDefault));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&f->d_mark), cap * sizeof(int)));
    HIPCHK(hipMemsetAsync(f->d_mark, 0, cap * sizeof(int), f->stream));
    f->rm_mark = 0;
    f->h_changed_cap = cap;
  }
  RmCand* hc = reinterpret_cast<RmCand*>(f->h_cand);
  for (size_t k = 0; k < nc; ++k) {
    hc[k].off = off[k], hc[k].start = start[k], hc[k].pad = 0u;
    f->h_changed[k] = 0;
  }
  f->h_changed[nc] = 0;  // (time-out word of k_rm_pool_bar's barrier: h_changed holds nc + nc / 2 + 64 entries)
  if (nc > RM_LDS) {  // too many candidates for the LDS table: the kernels search a device copy
    int rcs = frontier_ensure_stage(f, nc * sizeof(RmCand));
    if (rcs) return rcs;
    HIPCHK(hipMemcpyAsync(f->d_stage, hc, nc * sizeof(RmCand), hipMemcpyHostToDevice, f->stream));
    hc = reinterpret_cast<RmCand*>(f->d_stage);
  }
  if (nc <= RM_LDS && total <= RM_ONE_CELLS) {
    k_rm_pool_one<<<1, RM_ONE_T, 0, f->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, f->flag.p, f->pool, hc, (int)nc, total,
                                                 f->h_changed);
    FDBG("k_rm_pool_one");
    return FUELMI_OK;
  }
  const int mark = ++f->rm_mark;  // (marks of earlier searches never match: no clearing pass)
  if (nc <= RM_LDS && fblocks((long)total, 256) <= RM_BAR_BLOCKS) {
    if (!f->rm_bar) {
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&f->rm_bar), 64 * (RM_BAR_BLOCKS + 1)));
      HIPCHK(hipMemsetAsync(f->rm_bar, 0, 64 * (RM_BAR_BLOCKS + 1), f->stream));
      f->rm_bar_total = 0u;
    }
    const int nb = fblocks((long)total, 256);
    f->rm_bar_total += (u32)nb;
    k_rm_pool_bar<<<nb, 256, 0, f->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, f->flag.p, f->pool, hc, (int)nc, total, f->d_mark,
                                             mark, f->h_changed, f->rm_bar, f->rm_bar_total);
    FDBG("k_rm_pool_bar");
    return FUELMI_OK;
  }
  k_rm_pool<0><<<fblocks((long)total, 256), 256, 0, f->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, f->flag.p, f->pool, hc,
                                                               (int)nc, total, f->d_mark, mark, f->h_changed);
  FDBG("k_rm_pool<0>");
  k_rm_pool<1><<<fblocks((long)total, 256), 256, 0, f->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, f->flag.p, f->pool, hc,
                                                               (int)nc, total, f->d_mark, mark, f->h_changed);
  FDBG("k_rm_pool<1>");
  return FUELMI_OK;
}
// after the stream has drained: apply the verdicts.  removed_ids_ semantics (:74-85): index in
// frontiers_ as the list shrinks; dormant clusters are dropped silently.
static void remove_changed_end(fuelmi_frontier* f) {
  int erased_active = 0;
  if (!f->pend_rm.empty() && f->h_changed[f->pend_rm.size()] == -1) f->rm_failed = true;  // (k_rm_pool_bar's time-out)
  for (size_t k = 0; k < f->pend_rm.size(); ++k) {
    if (!f->h_changed[k]) continue;
    const fuelmi_frontier::PendingRm& p = f->pend_rm[k];
    if (p.list == &f->frontiers) {
      f->removed_ids.push_back(p.pos - erased_active);
      ++erased_active;
    }
    p.list->erase(p.it);
  }
  f->pend_rm.clear();
}

int frontier_regroup(fuelmi_frontier* f, const FArgs& F2, int npass) {
  const Geo& g = f->map->g;
  for (int p = 0; p < npass; ++p) {
    k_ms_hist<<<256, 256, 0, f->stream>>>(F2, p);
    k_ms_scan<<<1, 1024, 0, f->stream>>>(F2, p);
    k_ms_scatter<<<256, 256, 0, f->stream>>>(F2, p);
  }
  k_ms_info<<<256, 256, 0, f->stream>>>(g, F2);
  k_pack<<<1, 256, 0, f->stream>>>(F2);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

// the device pipeline of one search (all on f->stream; capturable)
static int frontier_enqueue_chain(fuelmi_frontier* f, int npass) {
  const Geo& g = f->map->g;
  FArgs& F = f->F;
  const int nb_max = (g.W + 255) / 256 + 1;  // surplus blocks exit on F.var->nblocks
  const int cgrid = 2048;
  // (the CURRENT buffer set's copy: a reset swaps F and F2, each with a device FVar of its own -- ADVICE r4)
  k_load_var<<<1, 64, 0, f->stream>>>(f->h_var, F.var_w);
  FDBG("k_load_var");
  k_pred<<<nb_max, 256, 0, f->stream>>>(g, F);
  FDBG("k_pred");
  k_scan_sums<<<1, 1024, 0, f->stream>>>(F);
  FDBG("k_scan_sums");
  k_compact<<<nb_max, 256, 0, f->stream>>>(g, F);
  FDBG("k_compact");
  if (f->ccl_tiles > 0) {
    // 512 threads per tile: the busiest tiles (a wall of ~2000 cells) set the kernel's duration, and their
    // cells are independent chains of LDS unions (measured 29.5 -> 24.1 us on G400; 1024 loses occupancy)
    k_ccl_local<512><<<f->ccl_tiles, 512, f->ccl_lds, f->stream>>>(g, F, f->TX, f->TY);
    FDBG("k_ccl_local");
    k_union<<<cgrid, 256, 0, f->stream>>>(g, F, f->TX, f->TY);
    FDBG("k_union");
  }
  k_claim<<<cgrid, 256, 0, f->stream>>>(g, F);
  FDBG("k_claim");
  k_sizes<<<cgrid, 256, 0, f->stream>>>(g, F);
  FDBG("k_sizes");  // grid-stride over 1024-cell chunks
  k_finalize<<<nb_max, 256, 0, f->stream>>>(g, F);
  FDBG("k_finalize");
  // ---- grouping + cluster info, still without touching the host ----
  k_rank_kept<<<16, 256, 0, f->stream>>>(F);
  FDBG("k_rank_kept");
  for (int p = 0; p < npass; ++p) {
    k_ms_hist<<<f->nb_launch, 256, 0, f->stream>>>(F, p);
    FDBG("k_ms_hist");
    k_ms_scan<<<1, 1024, 0, f->stream>>>(F, p);
    FDBG("k_ms_scan");
    k_ms_scatter<<<f->nb_launch, 256, 0, f->stream>>>(F, p);
    FDBG("k_ms_scatter");
  }
  k_ms_info<<<256, 256, 0, f->stream>>>(g, F);
  FDBG("k_ms_info");
  k_pack<<<1, 256, 0, f->stream>>>(F);
  FDBG("k_pack");
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

// FUELMI_HOST_TIMING: where the host's time inside _search_begin / _search_end goes (mean microseconds per call, printed
// by fuelmi_bench_cycles)
struct HostTiming {
  bool on = getenv("FUELMI_HOST_TIMING") != nullptr;
  double acc[16] = {0};
  long n = 0;
  std::chrono::steady_clock::time_point t;
  void start() {
    if (on) t = std::chrono::steady_clock::now(), ++n;
  }
  void lap(int k) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    acc[k] += std::chrono::duration<double, std::micro>(now - t).count();
    t = now;
  }
  void report() {
    if (!on || !n) return;
    static const char* names[16] = {"b:lists+reset", "b:box+wait+scope", "b:rm_begin", "b:region", "b:launch ccl", "b:events", "b:launch cross",
                                    "b:launch resolve", "b:launch out", "b:finish_reset", "e:pre-poll", "e:poll", "e:post-poll", "", "", ""};
    std::fprintf(stderr, "[host-timing] per search:");
    for (int k = 0; k < 13; ++k) std::fprintf(stderr, " %s %.2f", names[k], acc[k] / (double)n);
    std::fprintf(stderr, "\n");
    for (double& a : acc) a = 0;
    n = 0;
  }
};
static HostTiming g_ht;

// the fast chain: four direct launches (512-lane tile workgroups: 256 lanes measured 4-11 % slower in round 3 and again
// in round 6, profiles/r06_frontier_threads_prio_sweep.txt); falls back to the legacy one through counts[2] == 2
static int frontier_enqueue_fast(fuelmi_frontier* f) {
  const Geo& g = f->map->g;
  FArgs& F = f->F;
  // launch grid of the tile kernels: the chosen tile over the largest rectangle a search can cover
  const int qx = F.qbox.hi[0] - F.qbox.lo[0] + 2, qy = F.qbox.hi[1] - F.qbox.lo[1] + 2;
  const int mk = f->fast_menu;
  const int ftx = kFastMenu[mk][0], fty = kFastMenu[mk][1];
  const int tiles = ((qx + ftx - 1) / ftx) * ((qy + fty - 1) / fty);
  if (f->tl_ev) HIPCHK(hipEventRecord(f->tl_ev[0], f->stream));
  k_tile_ccl<512><<<tiles, 512, f->tile_lds[mk], f->stream>>>(g, F, *f->h_var);
  FDBG("k_tile_ccl");
  g_ht.lap(4);
  // the tile CCL (and the changed-cluster test in front of it) is the last reader of the map's occupancy planes: a
  // fusion queued behind this point may start as soon as it is done
  if (f->mark_planes_read) {
    HIPCHK(hipEventRecord(f->ev_planes_read, f->stream));
    map_add_plane_reader(f->map, f->ev_planes_read);
  } else {
    // nobody has rewritten the planes beside a running search of this finder yet (a plan cycle on a given map does
    // not; a streaming pipeline does from its first frame on): no barrier packet between the chain's kernels -- a
    // mutator that does arrive records the event behind what is queued and switches the marking on (map_wait_plane_readers)
    map_add_late_reader(f->map, f->stream, f->ev_planes_read, &f->mark_planes_read);
  }
  if (f->tl_ev) HIPCHK(hipEventRecord(f->tl_ev[1], f->stream));
  g_ht.lap(5);
  // round 6: the last workgroup of k_tile_cross resolves searches of up to 1 024 tile roots (FUELMI_FR_FUSE=0: never --
  // the A/B switch of profiles/r06_cross_resolve_fusion_ab.txt)
  static const bool fuse = !(getenv("FUELMI_FR_FUSE") && atoi(getenv("FUELMI_FR_FUSE")) == 0);
  {
    // (a finder whose last search had more tile roots than the launch's LDS holds -- the 800^2 x 200 map's full box: 2 465 --
    // does not try again for the next 16 searches: the attempt costs every workgroup a barrier and two atomics, -5 % there)
    bool fuse_now = fuse && f->h_var->ntiles_f > 0;  // (no tile, no last workgroup: k_resolve publishes the empty result)
    if (fuse_now && f->fuse_skip > 0) --f->fuse_skip, fuse_now = false;
    f->fuse_tried = fuse_now;
    const size_t lds_x = fuse_now ? std::max(f->cross_lds[mk], resolve_lds_bytes(1024)) : f->cross_lds[mk];
    const u32 rcap = fuse_now ? (u32)std::min<size_t>((lds_x - resolve_lds_bytes(0)) / 16, FR_RCAP) : 0u;
    // k_resolve is not even queued when this finder's last search was resolved in the launch with at most half the tile
    // roots the launch holds (a launch and a kernel boundary of the tail per search); a search that outgrows the guess
    // says so in its result (counts[2] == 3) and _search_end queues the two kernels then
    f->kr_queued = !(fuse_now && f->kr_skip_ok);
    k_tile_cross<512><<<tiles, 512, lds_x, f->stream>>>(g, F, rcap | (f->kr_queued ? 0u : 0x80000000u));
    f->rcap_used = rcap;
  }
  FDBG("k_tile_cross");
  g_ht.lap(6);
  if (f->kr_queued) k_resolve<<<1, RS_TK, f->resolve_lds, f->stream>>>(g, F);
  FDBG("k_resolve");
  g_ht.lap(7);
  if (f->tl_ev) HIPCHK(hipEventRecord(f->tl_ev[2], f->stream));
  k_tile_out<512><<<tiles, 512, f->out_lds[mk], f->stream>>>(g, F);
  FDBG("k_tile_out");
  g_ht.lap(8);
  if (f->tl_ev) HIPCHK(hipEventRecord(f->tl_ev[3], f->stream));
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

// fuelmi_frontier_reset, executed: frontier_flag_ becomes all-zero by SWAPPING to the spare plane (zeroed on a side
// stream since it was retired) -- no kernel of the search clears flags on the way, no clearing pass in front of it.
// The plane just retired is zeroed behind everything queued on the finder's stream so far (the tail of the last
// search still sets flags in it).  The kernel chains are captured once per plane (F.flag is a kernel argument).
// second half of frontier_apply_reset: queue the zeroing of the retired plane on the side stream.  Off the critical
// path: _search_begin calls it AFTER it has launched the new chain (three API calls the chain's start does not wait for)
static int frontier_finish_reset(fuelmi_frontier* f) {
  if (!f->zero_deferred) return FUELMI_OK;
  f->zero_deferred = false;
  const int W = f->map->g.W;
  // The plane just retired belongs to the stream just retired (they swap together), so the zeroing goes onto that
  // stream itself -- in order behind the tail that still sets flags in the plane and in front of the search after
  // next that uses it again.  One launch instead of wait + launch + record on a side stream (10.7 -> ~4 us of host
  // time per search, profiles/r05_host_timing.txt), and one stream fewer per finder (hardware queues).
  k_zero_words<<<fblocks(W, 256, 1024), 256, 0, f->stream2>>>(f->flag2.p, W);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}
static int frontier_apply_reset(fuelmi_frontier* f, bool defer_zeroing = false) {
  if (!f->fresh_pending) return FUELMI_OK;
  f->fresh_pending = false;
  // (everything that still writes the retired plane / the shared cell pool is in front of this; with two buffer sets only
  // a commit since the last swap makes anybody wait for it)
  if (f->pool_dirty) HIPCHK(hipEventRecord(f->ev_tail, f->stream));
  {
    // the other plane's buffer set and stream: the retiring search's tail keeps running on its own stream.  (The cell
    // pool is the one device buffer both sets share: if it was written since the last swap -- a commit -- the new
    // stream waits for the old one.)
    std::swap(f->stream, f->stream2);
    std::swap(f->F, f->F2);
    if (f->pool_dirty) HIPCHK(hipStreamWaitEvent(f->stream, f->ev_tail, 0));
    f->pool_dirty = false;
  }
  std::swap(f->flag, f->flag2);
  f->flag_cur ^= 1;
  f->F.flag = f->flag.p;
  f->zero_deferred = true;
  return defer_zeroing ? FUELMI_OK : frontier_finish_reset(f);
}

// searchFrontiers, first half: drops changed clusters and enqueues the whole device pipeline on the
// frontier's own stream (asynchronous).  The caller may queue other work of the cycle (inflation,
// ESDF, B-spline evaluation on the map's stream) before collecting the result with _search_end.
extern "C" int fuelmi_frontier_search_begin(fuelmi_frontier* f) {
  ARGCHK(f);
  FRONTIER_HAS_MAP(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_search_begin");
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  FArgs& F = f->F;
  f->tmp.clear();
  g_ht.start();
  {
    const int rcm = frontier_materialize_lists(f);  // (the result buffer is about to be reused)
    if (rcm) return rcm;
  }
  {
    const int rcr = frontier_apply_reset(f, true);
    if (rcr) return rcr;
  }
  struct FinishReset {  // (on every exit path, behind whatever was launched)
    fuelmi_frontier* f;
    ~FinishReset() {
      (void)frontier_finish_reset(f);
      g_ht.lap(9);
    }
  } finish_reset{f};
  g_ht.lap(0);
  double umin[3], umax[3];
  fuelmi_map_get_updated_box(m, umin, umax, 1);

  // the scan reads only the occupancy state planes: it runs on its own stream, ordered after the
  // last kernel that rewrote them (fusion / upload), and overlaps the inflation / ESDF / B-spline
  // kernels the caller has queued on the map's stream for the same cycle
  // (once per record and search stream: a plan cycle on an unchanged map queues no wait packet in front of its chain)
  if (f->planes_waited[f->flag_cur] != m->planes_ver + 1) {
    HIPCHK(hipStreamWaitEvent(f->stream, m->ev_planes, 0));
    f->planes_waited[f->flag_cur] = m->planes_ver + 1;
  }
  f->scope.reset(new StageScope(m, FUELMI_K_FRONTIER, f->stream));
  f->removed_ids.clear();
  for (int q = 0; q < 3; ++q) f->rm_lo[q] = 1, f->rm_hi[q] = 0;
  g_ht.lap(1);
  // on EVERY exit path below (the changed-cluster test reads the occupancy planes too, and an empty search or an
  // error returns before the chain): whatever was queued, the planes are free behind it
  struct PlanesRead {
    fuelmi_frontier* f;
    bool done = false;
    ~PlanesRead() {
      if (done || !f->map) return;
      if (!f->mark_planes_read)
        map_add_late_reader(f->map, f->stream, f->ev_planes_read, &f->mark_planes_read);
      else if (hipEventRecord(f->ev_planes_read, f->stream) == hipSuccess)
        map_add_plane_reader(f->map, f->ev_planes_read);
    }
  } planes_read{f};
  int rc = remove_changed_begin(f, umin, umax);
  if (rc) return rc;
  g_ht.lap(2);

  // scan box (:95-106): updated box +- (1,1,0.5) clipped to the exploration box, as indices
  const int nv[3] = {g.nx, g.ny, g.nz};
  FVar hv;
  memset(&hv, 0, sizeof(hv));
  bool empty = false;
  for (int k = 0; k < 3; ++k) {
    double infl = (k == 2) ? 0.5 : 1.0;
    double smin = std::max(umin[k] - infl, m->cfg.box_min[k]);
    double smax = std::min(umax[k] + infl, m->cfg.box_max[k]);
    int lo = (int)std::floor((smin - g.org[k]) * g.res_inv);
    int hi = (int)std::floor((smax - g.org[k]) * g.res_inv);
    lo = std::max(lo, 0);
    hi = std::min(hi, nv[k] - 1);  // the reference would index out of the map here (UB)
    hv.sbox.lo[k] = lo;
    hv.sbox.hi[k] = hi;
    if (lo > hi) empty = true;
  }
  f->pending = true;
  f->fusion_at_begin = m->fusion_count;
  f->search_empty = empty;
  f->fast_launched = false;
  if (empty) return FUELMI_OK;

  // Region that can hold Q0 cells: the scan box, the boxes of the clusters just dropped (their flags were
  // cleared), or the whole exploration box when flags / occupancy changed behind the updated-box
  // bookkeeping (fresh finder, reset, uploadOccupancy, resetBuffer).  Everything a previous search
  // could reach is flagged, and every later change of the occupancy lies inside this search's updated
  // box, so nothing outside the region can be grown into.
  int rlo[3], rhi[3];
  const bool all = f->dirty_all || f->seen_epoch != m->occ_epoch;
  for (int k = 0; k < 3; ++k) {
    rlo[k] = hv.sbox.lo[k], rhi[k] = hv.sbox.hi[k];
    if (f->rm_lo[0] <= f->rm_hi[0]) rlo[k] = std::min(rlo[k], f->rm_lo[k]), rhi[k] = std::max(rhi[k], f->rm_hi[k]);
    if (all) rlo[k] = 0, rhi[k] = nv[k] - 1;
    hv.qreg.lo[k] = std::max(F.qbox.lo[k], rlo[k]);
    hv.qreg.hi[k] = std::min(F.qbox.hi[k], rhi[k]);
  }
  f->dirty_all = true;  // until this search has completed
  bool have_q = true;
  for (int k = 0; k < 3; ++k)
    if (hv.qreg.lo[k] > hv.qreg.hi[k]) have_q = false;
  if (!have_q)
    for (int k = 0; k < 3; ++k) hv.qreg.lo[k] = 1, hv.qreg.hi[k] = 0;
  hv.nty = hv.ntiles = 0;
  hv.nty_f = hv.ntiles_f = 0;
  hv.ftx = 4, hv.fty = 8;
  if (have_q) {
    const int qx = hv.qreg.hi[0] - hv.qreg.lo[0] + 1, qy = hv.qreg.hi[1] - hv.qreg.lo[1] + 1;
    hv.nty = (qy + f->TY - 1) / f->TY;
    hv.ntiles = ((qx + f->TX - 1) / f->TX) * hv.nty;  // <= f->ccl_tiles (tiles of the whole Q box)
  }
  {
    // tiles of the fast chain: over the x/y bounding rectangle of the Q region and the scan box (NQ seeds start
    // clusters from the box_max face, one voxel outside the Q box; seeds are flagged wherever they are).
    // Tile of this search: 32 z-lines deep and 16 or 8 wide while that leaves >= 512 tiles (measured on full
    // 400^2 x 100 / 800^2 x 200 boxes: fewer tile roots and face pairs outweigh the longer tiles); smaller
    // tiles for small regions (a streaming search covers ~100 x 100 lines: a handful of big tiles would run
    // one after the other on a handful of CUs)
    int p0[2], p1[2];
    for (int k = 0; k < 2; ++k) {
      p0[k] = hv.sbox.lo[k], p1[k] = hv.sbox.hi[k];
      if (have_q) p0[k] = std::min(p0[k], hv.qreg.lo[k]), p1[k] = std::max(p1[k], hv.qreg.hi[k]);
      // (never beyond what the launch grids and per-tile tables were sized for)
      p0[k] = std::max(p0[k], F.qbox.lo[k]);
      p1[k] = std::min(p1[k], F.qbox.hi[k] + 1);
    }
    const int qx = p1[0] - p0[0] + 1, qy = p1[1] - p0[1] + 1;
    int pick = 3;
    for (int k = 0; k < 4; ++k)
      if (((qx + kFastMenu[k][0] - 1) / kFastMenu[k][0]) * ((qy + kFastMenu[k][1] - 1) / kFastMenu[k][1]) >= 512) {
        pick = k;
        break;
      }
    // (a finder whose tiles were too full lately starts on the smaller tile it ended on: the overflow costs a whole
    // chain run; forgotten after 64 searches)
    if (f->menu_min_ttl > 0) --f->menu_min_ttl, pick = std::max(pick, f->menu_min);
    f->fast_menu = pick;
    int ftx = kFastMenu[pick][0], fty = kFastMenu[pick][1];
    hv.ftx = ftx, hv.fty = fty;
    hv.px0 = p0[0], hv.py0 = p0[1], hv.px1 = p1[0], hv.py1 = p1[1];
    hv.ntx_f = std::max(0, (qx + ftx - 1) / ftx);
    hv.nty_f = std::max(0, (qy + fty - 1) / fty);
    hv.ntiles_f = hv.ntx_f * hv.nty_f;
  }
  // words to process: the x-slabs of the region plus one slab either side (neighbour look-ups of the
  // claims / unions read the Q0 plane there: it must not hold bits of an earlier search)
  auto adr = [&](const int* id) { return (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2]; };
  long a_lo = adr(hv.sbox.lo), a_hi = adr(hv.sbox.hi);
  if (have_q) {
    a_lo = std::min(a_lo, adr(hv.qreg.lo));
    a_hi = std::max(a_hi, adr(hv.qreg.hi));
  }
  a_lo = std::max(0L, a_lo - 2L * g.nyz);
  a_hi = std::min((long)g.N - 1, a_hi + 2L * g.nyz);
  hv.w0 = (int)((a_lo >> 6) & ~255L);
  const int w_hi = (int)(a_hi >> 6);
  hv.nblocks = (w_hi - hv.w0) / 256 + 1;
  hv.nwords = hv.nblocks * 256;
  const bool fast = f->fast_ok;
  hv.fresh = 0;  // (a reset has already swapped in an all-zero flag plane, frontier_apply_reset)
  if (F.dbg)  // FUELMI_FR_TIMING: stamps of this search only
    HIPCHK(hipMemsetAsync(F.dbg, 0, (size_t)(3 * (hv.ntiles_f + 1)) * FR_DBG_SLOTS * sizeof(unsigned long long), f->stream));
  hv.epoch = ++f->epoch;
  if (hv.epoch == 0u) hv.epoch = f->epoch = 1u;
  // pinned; read by the first kernel of the chain.  (The previous search's first kernel ran long ago -- its
  // result was collected -- and the tail that may still be running reads the device copy.)
  *f->h_var = hv;

  // the second radix pass is needed only beyond 256 kept clusters: guess from the previous search
  f->npass = f->last_nkept > 192 ? 2 : 1;
  f->nb_launch = 256;  // the multisplit kernels stride over however many 2048-cell chunks there are

  // The fast chain's four kernels are launched DIRECTLY (rounds 2-3 replayed them as a hipGraph: cheaper for the host, but
  // the graph reaches the device ~10 us later, which is on the cycle's critical path since a fresh search runs beside the
  // previous one's tail -- same-box A/B in profiles/r04_tuning_ab_cycle.txt --, and only direct launches can mark the
  // point behind which the next depth frame may be fused).  The legacy chain is ~23 dependent launches whose arguments
  // never change (everything per-search sits behind F.var): it is replayed as a hipGraph.
  static const bool no_graph = getenv("FUELMI_DEBUG_SYNC") != nullptr;
  g_ht.lap(3);
  if (fast) {
    f->fast_launched = true;
    planes_read.done = true;
    return frontier_enqueue_fast(f);
  }
  if (no_graph) return frontier_enqueue_chain(f, f->npass);
  hipGraphExec_t& exec = f->graph_exec[f->npass - 1][f->flag_cur];
  if (!exec) {
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(f->stream, hipStreamCaptureModeThreadLocal));
    const int rc2 = frontier_enqueue_chain(f, f->npass);
    const hipError_t ec = hipStreamEndCapture(f->stream, &graph);
    if (rc2) return rc2;
    HIPCHK(ec);
    HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(graph));
  }
  HIPCHK(hipGraphLaunch(exec, f->stream));
  return FUELMI_OK;
}

// searchFrontiers, second half: waits for the pipeline and assembles tmp_frontiers_.
extern "C" int fuelmi_frontier_search_end(fuelmi_frontier* f, int* n_new) {
  ARGCHK(f && n_new);
  *n_new = 0;
  FRONTIER_HAS_MAP(f);
  if (!f->pending) {
    fuelmi_set_error("fuelmi_frontier_search_end without a matching _begin");
    return FUELMI_EINVAL;
  }
  f->pending = false;
  struct ScopeEnd {  // closes the profiling bracket on every exit path
    fuelmi_frontier* f;
    ~ScopeEnd() { f->scope.reset(); }
  } scope_end{f};
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  // (a plane mutator issued from here on comes behind this call's wait for the chain's result: the plane-reading
  // kernels are done by then -- nothing left for it to wait for)
  struct DropLate {
    fuelmi_frontier* f;
    ~DropLate() {
      if (f->map) map_drop_late_reader(f->map, f->ev_planes_read);
    }
  } drop_late{f};
  if (f->search_empty) {
    if (!f->pend_rm.empty()) {
      HIPCHK(hipStreamSynchronize(f->stream));
      remove_changed_end(f);
    }
    return FUELMI_OK;
  }
  const Geo& g = m->g;
  FArgs& F = f->F;
  const int nb_launch = f->nb_launch;
  u32* counts = F.h_counts;
  KeptRec* h_rec = F.h_rec;
  const u32* h_part = F.h_part;
  const u32* h_cells = F.h_cells;
  // poll instead of a blocking wait: the caller is about to consume the result and the chain is
  // ~100 us long, an interrupt-driven wake-up costs a noticeable fraction of that
  F.fast = 0;
  g_ht.start();
  if (g_ht.on) --g_ht.n;
  g_ht.lap(10);
  if (f->fast_launched) {
    // the fast chain publishes counts + cluster records from k_resolve and stamps them with the search's epoch;
    // the two kernels behind it (flags, regrouping + copy-out of the cells) keep running -- whoever touches the
    // cell lists waits for them (frontier_tail_sync)
    volatile u32* stamp = counts + 15;
    const u32 want = f->h_var->epoch;
    unsigned spins = 0;
    const auto w0 = std::chrono::steady_clock::now();
    struct WaitAcc {  // (time spent polling: the part of _search_end that is the device's, not the host's)
      fuelmi_frontier* f;
      std::chrono::steady_clock::time_point w0;
      ~WaitAcc() { f->wait_us_acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count(); }
    };
    {
    WaitAcc wacc{f, w0};
    const bool yld = poll_yields();
    while (*stamp != want) {
      if (yld) std::this_thread::yield();
      if ((++spins & 0x3FFFu) == 0u) {  // every ~16k polls: has the stream died under us?
        const hipError_t q = hipStreamQuery(f->stream);
        if (q != hipErrorNotReady && q != hipSuccess) HIPCHK(q);
        if (q == hipSuccess && *stamp != want) {
          fuelmi_set_error("frontier search: the chain finished without publishing its result");
          return FUELMI_EHIP;
        }
      }
    }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    g_ht.lap(11);
    f->tail_pending = true;
    F.fast = 1;
    if (F.dbg) {  // FUELMI_FR_TIMING: where the tile kernel and the resolve kernel spend their time (100 MHz ticks)
      (void)hipStreamSynchronize(f->stream);
      const int nt = f->h_var->ntiles_f;
      std::vector<unsigned long long> d((size_t)(3 * (nt + 1)) * FR_DBG_SLOTS);
      (void)hipMemcpy(d.data(), F.dbg, d.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull, t1 = 0;
      double phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int busiest = 0, nonempty = 0;
      for (int b = 0; b < nt; ++b) {
        const unsigned long long* r = &d[(size_t)b * FR_DBG_SLOTS];
        if (!r[0]) continue;
        t0 = std::min(t0, r[0]);
        for (int k = 1; k < 8; ++k)
          if (r[k]) t1 = std::max(t1, r[k]);
        if (r[8] > d[(size_t)busiest * FR_DBG_SLOTS + 8]) busiest = b;
        if (r[8]) ++nonempty;
      }
      {
        double st_av = 0, st_mx = 0, du_av = 0, du_mx = 0;
        int cnt = 0;
        for (int b = 0; b < nt; ++b) {
          const unsigned long long* r = &d[(size_t)b * FR_DBG_SLOTS];
          if (!r[0] || !r[6] || r[6] < r[0]) continue;
          const double st = (double)(r[0] - t0) / 100.0, du = (double)(r[6] - r[0]) / 100.0;
          st_av += st, st_mx = std::max(st_mx, st), du_av += du, du_mx = std::max(du_mx, du), ++cnt;
        }
        std::fprintf(stderr, "[fr-timing] ccl: %d tiles with cells: start avg %.1f max %.1f, duration avg %.1f max %.1f us\n", cnt,
                     st_av / std::max(cnt, 1), st_mx, du_av / std::max(cnt, 1), du_mx);
      }
      const unsigned long long* rb = &d[(size_t)busiest * FR_DBG_SLOTS];
      for (int k = 1; k < 8; ++k) phase[k] = rb[k] && rb[k - 1] ? (double)(rb[k] - rb[k - 1]) / 100.0 : 0.0;
      std::fprintf(stderr, "[fr-timing] tiles %d (non-empty %d) span %.1f us; busiest tile %d (%llu cells, start +%.1f us): "
                   "bits %.1f labels %.1f unions %.1f roots %.1f records %.1f write %.1f us; %llu unions, %llu find steps, %llu links, %llu roots\n", nt, nonempty,
                   (double)(t1 - t0) / 100.0, busiest, rb[8], (double)(rb[0] - t0) / 100.0, phase[1], phase[2], phase[3],
                   phase[4], phase[5], phase[6], rb[9], rb[10], rb[11], rb[12]);
      for (int kern = 1; kern <= 2; ++kern) {  // k_tile_cross, k_tile_out: rows behind those of the tiles and of k_resolve
        const int np = kern == 1 ? 4 : 6;
        double mx[8] = {0}, av[8] = {0};
        unsigned long long c0 = ~0ull, c1 = 0;
        int cnt = 0;
        for (int b = 0; b < nt; ++b) {
          const unsigned long long* r = &d[(size_t)(kern * (nt + 1) + b) * FR_DBG_SLOTS];
          if (!r[0] || !r[np]) continue;
          c0 = std::min(c0, r[0]), c1 = std::max(c1, r[np]);
          for (int k = 1; k <= np; ++k) {
            const double ph = r[k] && r[k - 1] ? (double)(r[k] - r[k - 1]) / 100.0 : 0.0;
            mx[k] = std::max(mx[k], ph), av[k] += ph;
          }
          ++cnt;
        }
        double st_av = 0, st_mx = 0, du_av = 0, du_mx = 0;
        for (int b = 0; b < nt; ++b) {
          const unsigned long long* r = &d[(size_t)(kern * (nt + 1) + b) * FR_DBG_SLOTS];
          if (!r[0] || !r[np]) continue;
          const double st = (double)(r[0] - c0) / 100.0, du = (double)(r[np] - r[0]) / 100.0;
          st_av += st, st_mx = std::max(st_mx, st), du_av += du, du_mx = std::max(du_mx, du);
        }
        {
          unsigned long long e0 = ~0ull;
          double ea = 0, em = 0, la = 0, lm = 0;
          int n2 = 0;
          for (int b = 0; b < nt; ++b) {
            const unsigned long long v = d[(size_t)b * FR_DBG_SLOTS + 12 + kern];
            if (v) e0 = std::min(e0, v);
          }
          for (int b = 0; b < nt; ++b) {
            const unsigned long long v = d[(size_t)b * FR_DBG_SLOTS + 12 + kern];
            const unsigned long long* r = &d[(size_t)(kern * (nt + 1) + b) * FR_DBG_SLOTS];
            if (!v || !r[0] || r[0] < v) continue;
            const double e = (double)(v - e0) / 100.0, l = (double)(r[0] - v) / 100.0;
            ea += e, em = std::max(em, e), la += l, lm = std::max(lm, l), ++n2;
          }
          std::fprintf(stderr, "[fr-timing]   entry avg %.1f max %.1f us after the first workgroup; entry -> first stamp avg %.1f max %.1f us\n",
                       ea / std::max(n2, 1), em, la / std::max(n2, 1), lm);
        }
        std::fprintf(stderr, "[fr-timing] %s: span %.1f us over %d tiles; start avg %.1f max %.1f, duration avg %.1f max %.1f; phases avg/max:",
                     kern == 1 ? "cross (bits, build, list, append)" : "out (bits, cells, column, first, rows, flags)",
                     cnt ? (double)(c1 - c0) / 100.0 : 0.0, cnt, st_av / std::max(cnt, 1), st_mx, du_av / std::max(cnt, 1), du_mx);
        for (int k = 1; k <= np; ++k) std::fprintf(stderr, " %.1f/%.1f", av[k] / std::max(cnt, 1), mx[k]);
        std::fprintf(stderr, "\n");
      }
      const unsigned long long* rr = &d[(size_t)nt * FR_DBG_SLOTS];
      std::fprintf(stderr, "[fr-timing] resolve: init %.1f unions %.1f find %.1f reduce %.1f clusters %.1f seeds %.1f rank %.1f "
                   "codes %.1f publish %.1f us; pairs %llu roots %llu cells %llu\n", (rr[1] - rr[0]) / 100.0,
                   (rr[2] - rr[1]) / 100.0, (rr[3] - rr[2]) / 100.0, (rr[4] - rr[3]) / 100.0, (rr[5] - rr[4]) / 100.0, 0.0,
                   (rr[6] - rr[5]) / 100.0, (rr[7] - rr[6]) / 100.0, (rr[8] - rr[7]) / 100.0, rr[9], rr[10], rr[11]);
    }
    if (counts[2] == 3u) {  // the search outgrew the launch that was to resolve it and no k_resolve was queued: do it now
      const int mk = f->fast_menu;
      const int qx = F.qbox.hi[0] - F.qbox.lo[0] + 2, qy = F.qbox.hi[1] - F.qbox.lo[1] + 2;
      const int tiles = ((qx + kFastMenu[mk][0] - 1) / kFastMenu[mk][0]) * ((qy + kFastMenu[mk][1] - 1) / kFastMenu[mk][1]);
      *stamp = 0u;
      k_resolve<<<1, RS_TK, f->resolve_lds, f->stream>>>(g, F);
      k_tile_out<512><<<tiles, 512, f->out_lds[mk], f->stream>>>(g, F);
      HIPCHK(hipGetLastError());
      unsigned spins2 = 0;
      while (*stamp != want) {
        if ((++spins2 & 0x3FFFu) == 0u) {
          const hipError_t q = hipStreamQuery(f->stream);
          if (q != hipErrorNotReady && q != hipSuccess) HIPCHK(q);
          if (q == hipSuccess && *stamp != want) {
            fuelmi_set_error("frontier search: k_resolve finished without publishing its result");
            return FUELMI_EHIP;
          }
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      ++f->n_late_resolve;
    }
    f->kr_skip_ok = counts[2] == 0u && counts[9] == 1u && 2u * counts[10] <= f->rcap_used;
    ++f->n_fast;
    if (counts[9] == 1u)
      ++f->n_in_launch;
    else if (f->fuse_tried && counts[2] == 0u)
      f->fuse_skip = 16;  // (too many tile roots for the launch's LDS: k_resolve did it)
    if (counts[2] == 2u && m->fusion_count != f->fusion_at_begin) {
      fuelmi_set_error("frontier search: the fast chain overflowed AFTER the map was fused again (a frame queued between "
                       "_search_begin and _search_end): the occupancy this search was about is gone -- call _search_end before the next fusion "
                       "on inputs this noisy");
      // the changed clusters' flags are already cleared on the device: take them off the lists too, and make the next
      // search look at the whole box again (the lists and the flag plane must not part ways on this path)
      remove_changed_end(f);
      f->dirty_all = true;
      return FUELMI_ELIMIT;
    }
    // A capacity of ONE TILE was exceeded (codes 11 cells, 12 components, 15 root pairs, 18 work-list entries -- a
    // full-height frontier wall along a y-line puts 32 x nz cells into one 8 x 32 tile): nothing was modified, and
    // the menu's next, smaller tile holds half as much per tile -- run the chain again on it before giving the search to
    // the legacy chain (round 6, with FR_TCELL = 2 048).
    while (counts[2] == 2u && f->fast_menu < 3 && (counts[6] == 11u || counts[6] == 12u || counts[6] == 15u || counts[6] == 18u)) {
      HIPCHK(frontier_tail_sync(f) == FUELMI_OK ? hipSuccess : hipErrorUnknown);
      ++f->fast_menu;
      FVar hv = *f->h_var;
      const int qx = hv.px1 - hv.px0 + 1, qy = hv.py1 - hv.py0 + 1;
      hv.ftx = kFastMenu[f->fast_menu][0], hv.fty = kFastMenu[f->fast_menu][1];
      hv.ntx_f = std::max(0, (qx + hv.ftx - 1) / hv.ftx);
      hv.nty_f = std::max(0, (qy + hv.fty - 1) / hv.fty);
      hv.ntiles_f = hv.ntx_f * hv.nty_f;
      hv.epoch = ++f->epoch;
      if (hv.epoch == 0u) hv.epoch = f->epoch = 1u;
      *f->h_var = hv;
      f->kr_skip_ok = false;  // (k_resolve queued: no second round trip whatever the root count)
      if (getenv("FUELMI_FR_TIMING")) std::fprintf(stderr, "[fr-timing] capacity code %u: again with %d x %d tiles\n", counts[6], hv.ftx, hv.fty);
      const int rcr = frontier_enqueue_fast(f);
      if (rcr) return rcr;
      unsigned spins3 = 0;
      while (*stamp != hv.epoch) {
        if ((++spins3 & 0x3FFFu) == 0u) {
          const hipError_t q = hipStreamQuery(f->stream);
          if (q != hipErrorNotReady && q != hipSuccess) HIPCHK(q);
          if (q == hipSuccess && *stamp != hv.epoch) {
            fuelmi_set_error("frontier search: the re-tiled chain finished without publishing its result");
            return FUELMI_EHIP;
          }
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      f->tail_pending = true;
      ++f->n_retiled;
      f->menu_min = f->fast_menu, f->menu_min_ttl = 64;
    }
    if (counts[2] == 2u) {
      // a capacity of the fast path was exceeded (noise-like input): nothing was modified; run the legacy chain
      --f->n_fast;
      ++f->n_fallback;
      if (getenv("FUELMI_FR_TIMING")) std::fprintf(stderr, "[fr-timing] fallback to the legacy chain: capacity code %u\n", counts[6]);
      HIPCHK(frontier_tail_sync(f) == FUELMI_OK ? hipSuccess : hipErrorUnknown);
      F.fast = 0;
      f->npass = 1;
      int rcl = frontier_enqueue_chain(f, f->npass);
      if (rcl) return rcl;
      HIPCHK(hipStreamSynchronize(f->stream));
    }
  } else {
    ++f->n_legacy;
    hipError_t q;
    while ((q = hipStreamQuery(f->stream)) == hipErrorNotReady) {
    }
    HIPCHK(q);
  }
  remove_changed_end(f);
  if (f->rm_failed) {
    f->rm_failed = false;
    f->dirty_all = true;
    fuelmi_set_error("frontier search: the changed-cluster test's in-kernel barrier timed out (device oversubscribed for 50 ms?)");
    return FUELMI_EHIP;
  }
  if (counts[2] || counts[3] > F.cap_kept) {
    fuelmi_set_error("frontier capacity exceeded (cells %u/%u seeds %u/%u clusters %u/%u)", counts[0], F.cap_q,
                     counts[1], F.cap_s, counts[3], F.cap_kept);
    return FUELMI_ELIMIT;
  }
  const u32 nq = counts[0], nkept = counts[3], n_out = counts[5];
  f->last_nkept = (int)nkept;
  if (nkept > 256 && f->npass < 2) {
    // more than 256 clusters but only one radix pass was enqueued: run the high-digit pass now
    for (int p = 1; p < 2; ++p) {
      k_ms_hist<<<nb_launch, 256, 0, f->stream>>>(F, p);
      FDBG("k_ms_hist");
      k_ms_scan<<<1, 1024, 0, f->stream>>>(F, p);
      FDBG("k_ms_scan");
      k_ms_scatter<<<nb_launch, 256, 0, f->stream>>>(F, p);
      FDBG("k_ms_scatter");
    }
    // cluster records must be re-initialised before the accumulators are refilled
    k_rank_kept<<<16, 256, 0, f->stream>>>(F);
    FDBG("k_rank_kept");
    k_ms_info<<<256, 256, 0, f->stream>>>(g, F);
    FDBG("k_ms_info");
    k_pack<<<1, 256, 0, f->stream>>>(F);
    FDBG("k_pack");
    HIPCHK(hipStreamSynchronize(f->stream));
  }
  if (nkept == 0) {
    f->dirty_all = false;
    f->seen_epoch = m->occ_epoch;
    return FUELMI_OK;
  }
  u32 ncl = nkept, ncells = n_out;
  std::vector<std::vector<float>> filtered;
  const bool split_mode = f->cfg.split != 0;
  // reference_order 1: always; 2 ("auto", the facade's default): whenever every cluster of the search holds at most
  // FR_REFORDER_AUTO cells (the level sweep then runs inside LDS) -- every incremental search of an exploration
  // run -- and the canonical order for the giant ones
  bool ref_order = f->cfg.reference_order == 1;
  if (f->cfg.reference_order == 2) {
    ref_order = true;
    for (u32 r = 0; r < nkept; ++r)
      if (h_rec[r].size > FR_REFORDER_AUTO) ref_order = false;
  }
  f->ref_now = ref_order;
  // never silent: which order this search delivers is part of its result -- fuelmi_frontier_order_stats
  f->order_last = ref_order ? 1 : 0;
  if (ref_order)
    ++f->n_order_ref;
  else if (f->cfg.reference_order == 2) {
    ++f->n_order_fallback;
    u32 big = 0;
    for (u32 r = 0; r < nkept; ++r) big = std::max(big, h_rec[r].size);
    f->order_fallback_cells = big;
  }
  int fin = nkept <= 256 ? 1 : 0;  // buffer pair holding the grouped cells of the search
  std::vector<u32> off2;
  u32 n_in = n_out;
  if (split_mode && !ref_order) HIPCHK(frontier_tail_sync(f) == FUELMI_OK ? hipSuccess : hipErrorUnknown);
  if (ref_order) {  // cells of every cluster in expandFrontier's order (an NQ seed first), into the other pair
    int rc = frontier_reference_order(f, nq, nkept, n_out, fin, &n_in, &off2, !split_mode);  // (queued behind the
    if (rc) return rc;                                                // search's tail; returns with the stream idle)
    f->tail_pending = false;
    fin = 1 - fin;
    ncells = n_in;
  }
  if (split_mode) {  // splitLargeFrontiers (:120) on the device; the pinned buffers then hold the pieces
    int rc = frontier_split_run(f, nq, nkept, n_in, fin, &ncl, &ncells, &filtered);
    if (rc) return rc;
    fin = ncl <= 256 ? 1 : 0;  // where the regrouping of the pieces ended
  }  // (reference order without the split: the ordered cells are in h_cells already)
  f->last_fin = fin;  // buffer holding the grouped cells the lazy clusters point into
  f->cells_fetch = F.fast && !ref_order && !split_mode && nq > F.hcells_direct_max;
  f->cells_fetch_n = ncells;
  // (the fast chain's records are complete; the legacy chain and the regrouping of the split pieces leave
  // per-chunk partial records to fold)
  const u32 nchunk = ((ref_order || F.fast) && !split_mode) ? 0u : (ncells + SZ_CH - 1) / SZ_CH;  // (records of the grouped
                                                                                       // array, not of the ordered one)
  for (u32 c = 0; c < nchunk; ++c) {  // fold the per-chunk records into the per-cluster totals
    const u32* rec = h_part + (size_t)c * 10;
    const u32 r = rec[0];
    if (r >= ncl) continue;
    for (int q = 0; q < 3; ++q) {
      h_rec[r].sum[q] += rec[1 + q];
      h_rec[r].box[q] = std::min(h_rec[r].box[q], rec[4 + q]);
      h_rec[r].box[3 + q] = std::max(h_rec[r].box[3 + q], rec[7 + q]);
    }
  }

  // host: bulk copies only (+ inserting an NQ seed where a seed started the cluster); records are
  // already in creation order (ascending claimer address = the reference's scan order)
  for (u32 r = 0; r < ncl; ++r) {
    const KeptRec& kr = h_rec[r];
    f->tmp.emplace_back();
    HCluster& c = f->tmp.back();
    const bool seed = !split_mode && kr.slot >= nq;
    const u32 cnt = kr.size - (seed ? 1u : 0u);
    c.lazy = reinterpret_cast<const int*>(h_cells) + kr.off;
    c.lazy_n = cnt;
    if (ref_order && !split_mode) {  // ordered list: the seed (if any) is already its first cell
      c.lazy = reinterpret_cast<const int*>(h_cells) + off2[r];
      c.lazy_n = kr.size;
    }
    unsigned long long sum[3] = {kr.sum[0], kr.sum[1], kr.sum[2]};
    u32 lo[3] = {kr.box[0], kr.box[1], kr.box[2]};
    u32 hi[3] = {kr.box[3], kr.box[4], kr.box[5]};
    if (split_mode) {
      // an NQ seed travels as the LAST cell of its piece (already in the sums): restore address order
      if (!ref_order && cnt > 1 && c.lazy[cnt - 1] < c.lazy[cnt - 2]) {
        c.lazy_seed = c.lazy[cnt - 1];
        c.lazy_n = cnt - 1;
      }
      c.filtered.swap(filtered[r]);
    }
    if (seed) {
      const int a = (int)kr.addr;
      if (!ref_order) c.lazy_seed = a;
      const u32 x = (u32)a / (u32)g.nyz, rr = (u32)a - x * (u32)g.nyz, y = rr / (u32)g.nz, z = rr - y * (u32)g.nz;
      const u32 id[3] = {x, y, z};
      for (int q = 0; q < 3; ++q) {
        sum[q] += id[q];
        lo[q] = cnt ? std::min(lo[q], id[q]) : id[q];
        hi[q] = cnt ? std::max(hi[q], id[q]) : id[q];
      }
    }
    // computeFrontierInfo (:374-390): mean of voxel centres = centre of the mean index
    const double nn = (double)c.size();
    for (int q = 0; q < 3; ++q) {
      c.avg[q] = ((double)sum[q] / nn + 0.5) * g.res + g.org[q];
      c.bmin[q] = ((int)lo[q] + 0.5) * g.res + g.org[q];
      c.bmax[q] = ((int)hi[q] + 0.5) * g.res + g.org[q];
    }
    if (ref_order) {
      // the reference's own evaluation: cell centres added one by one in list order, then divided; the box
      // from the same centres (identical to the index form above)
      double sm[3] = {0.0, 0.0, 0.0};
      u32 blo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bhi[3] = {0u, 0u, 0u};
      const int* cl = c.lazy;
      const u32 ncell = c.lazy_n;
      for (u32 i = 0; i < ncell; ++i) {
        const u32 a = (u32)cl[i];
        const u32 x = a / (u32)g.nyz, rr = a - x * (u32)g.nyz, y = rr / (u32)g.nz, z = rr - y * (u32)g.nz;
        const u32 id[3] = {x, y, z};
        for (int q = 0; q < 3; ++q) {
          sm[q] += ((double)id[q] + 0.5) * g.res + g.org[q];
          blo[q] = std::min(blo[q], id[q]);
          bhi[q] = std::max(bhi[q], id[q]);
        }
      }
      for (int q = 0; q < 3; ++q) {
        c.avg[q] = sm[q] / (double)ncell;
        c.bmin[q] = ((int)blo[q] + 0.5) * g.res + g.org[q];
        c.bmax[q] = ((int)bhi[q] + 0.5) * g.res + g.org[q];
      }
    }
  }
  *n_new = (int)f->tmp.size();
  f->dirty_all = false;
  f->seen_epoch = m->occ_epoch;
  g_ht.lap(12);
  return FUELMI_OK;
}


extern "C" int fuelmi_frontier_search(fuelmi_frontier* f, int* n_new) {
  ARGCHK(f && n_new);
  int rc = fuelmi_frontier_search_begin(f);
  if (rc) return rc;
  return fuelmi_frontier_search_end(f, n_new);
}

extern "C" int fuelmi_frontier_synchronize(fuelmi_frontier* f) {
  ARGCHK(f);
  FRONTIER_HAS_MAP(f);
  HIPCHK(hipSetDevice(f->device));
  f->tail_pending = false;
  HIPCHK(frontier_drain(f));
  return FUELMI_OK;
}
extern "C" int fuelmi_frontier_resolved_in_launch(const fuelmi_frontier* f) { return f ? f->n_in_launch : FUELMI_EINVAL; }
extern "C" int fuelmi_frontier_stats(const fuelmi_frontier* f, int out3[3]) {
  ARGCHK(f && out3);
  out3[0] = f->n_fast, out3[1] = f->n_legacy, out3[2] = f->n_fallback;
  return FUELMI_OK;
}
extern "C" int fuelmi_frontier_order_stats(const fuelmi_frontier* f, int out4[4]) {
  ARGCHK(f && out4);
  out4[0] = f->order_last, out4[1] = f->n_order_ref, out4[2] = f->n_order_fallback, out4[3] = (int)f->order_fallback_cells;
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_reset(fuelmi_frontier* f) {
  ARGCHK(f);
  FRONTIER_HAS_MAP(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_reset");
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  f->frontiers.clear();
  f->dormant.clear();
  f->prev.clear();
  if (f->keep_prev && f->stream2 && !f->fresh_pending && !f->tmp.empty()) {
    // the new clusters of the last search stay readable as list 3 (see fuelmi_frontier::prev): their cells are in
    // the buffer set this reset retires.  Whatever still has to happen to them -- the search's tail, the copy of a
    // large grouped cell list to the pinned block -- is queued on the retiring stream now and waited for by whoever
    // reads list 3, not here
    f->prev.swap(f->tmp);
    if (f->cells_fetch) {
      f->cells_fetch = false;
      HIPCHK(hipMemcpyAsync(f->F.h_cells, f->F.ms_val[f->last_fin], (size_t)f->cells_fetch_n * sizeof(u32), hipMemcpyDeviceToHost,
                            f->stream));
    }
    HIPCHK(hipEventRecord(f->ev_prev, f->stream));
    f->prev_pending = true;
    f->tail_pending = false;
  }
  f->tmp.clear();
  f->removed_ids.clear();
  f->dirty_all = true;
  f->pool_used = 0;
  f->fresh_pending = true;  // executed by the next search (folded into its first kernel) or by whoever reads the flags
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_keep_previous(fuelmi_frontier* f, int on) {
  ARGCHK(f);
  f->keep_prev = on != 0;
  if (!on) f->prev.clear();
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_commit(fuelmi_frontier* f, int dormant) {
  ARGCHK(f);
  FRONTIER_HAS_MAP(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_commit");
  auto& dst = dormant ? f->dormant : f->frontiers;
  HIPCHK(hipSetDevice(f->map->device));
  const int rc = frontier_keep_clusters(f, f->tmp);
  if (rc) return rc;
  dst.splice(dst.end(), f->tmp);
  return FUELMI_OK;
}

static const std::list<HCluster>* pick(const fuelmi_frontier* f, int which) {
  return which == 0 ? &f->tmp : (which == 1 ? &f->frontiers : (which == 2 ? &f->dormant : (which == 3 ? &f->prev : nullptr)));
}
static const HCluster* nth(const fuelmi_frontier* f, int which, int k) {
  const std::list<HCluster>* L = pick(f, which);
  if (!L || k < 0 || k >= (int)L->size()) return nullptr;
  auto it = L->begin();
  std::advance(it, k);
  return &*it;
}
extern "C" int fuelmi_frontier_count(const fuelmi_frontier* f, int which) {
  ARGCHK(f && pick(f, which));
  return (int)pick(f, which)->size();
}
extern "C" int fuelmi_frontier_cluster_size(const fuelmi_frontier* f, int which, int k) {
  ARGCHK(f);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  return (int)c->size();
}
extern "C" int fuelmi_frontier_cluster_cells(const fuelmi_frontier* f, int which, int k, int* adr) {
  ARGCHK(f && adr);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  if (c->lazy) {
    int rc = which == 3 ? frontier_prev_ready(f) : frontier_cells_ready(f);
    if (rc) return rc;
  }
  {
    const int rcf = frontier_fetch_cluster(f, c);
    if (rcf) return rcf;
  }
  c->copy_to(adr);
  return FUELMI_OK;
}
// A few helper threads for the host-side bulk steps of result delivery (decoding a large cluster's cells): created on
// first use, parked on a condition variable, shared by every finder of the process.  FUELMI_HOST_HELPERS=0 switches them
// off (everything then runs on the calling thread).
namespace {
struct HostHelpers {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  const std::function<void(size_t, size_t)>* job = nullptr;
  size_t n = 0, piece = 0;
  std::atomic<size_t> next{0};
  int gen = 0, running = 0;
  bool stop = false;
  void worker() {
    int seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
      }
      take();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--running == 0) cv_done.notify_one();
      }
    }
  }
  void take() {
    for (;;) {
      const size_t i0 = next.fetch_add(piece);
      if (i0 >= n) return;
      (*job)(i0, std::min(n, i0 + piece));
    }
  }
  void run(size_t total, const std::function<void(size_t, size_t)>& f) {
    static const int want = getenv("FUELMI_HOST_HELPERS") ? atoi(getenv("FUELMI_HOST_HELPERS")) : 3;
    if (want <= 0) {
      f(0, total);
      return;
    }
    std::unique_lock<std::mutex> lk(mu);
    if (th.empty())
      for (int k = 0; k < want; ++k) th.emplace_back([this] { worker(); });
    job = &f, n = total, piece = std::max<size_t>(8192, (total + 4 * (th.size() + 1) - 1) / (4 * (th.size() + 1)));
    next = 0;
    running = (int)th.size();
    ++gen;
    lk.unlock();
    cv_go.notify_all();
    take();  // the caller works too
    lk.lock();
    cv_done.wait(lk, [&] { return running == 0; });
  }
  ~HostHelpers() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_go.notify_all();
    for (auto& t : th) t.join();
  }
};
std::mutex g_helpers_once;
}  // namespace
static void host_parallel_for(size_t n, const std::function<void(size_t, size_t)>& f) {
  static HostHelpers* H = new HostHelpers;  // (leaked on purpose: joining threads from a static destructor at exit can hang)
  std::lock_guard<std::mutex> lk(g_helpers_once);  // one bulk step at a time
  H->run(n, f);
}

// Frontier::cells_ as the reference's callers hold them: the voxel CENTRES of cluster k, three doubles per cell (the
// layout of a vector<Eigen::Vector3d>), decoded by the library straight out of the pinned result block into the
// caller's storage -- no intermediate address list, and no division per cell: consecutive cells of a list mostly share
// a z-line (ascending addresses) or neighbour one (BFS order), so the line's base address is carried along.
extern "C" int fuelmi_frontier_cluster_centres(const fuelmi_frontier* f, int which, int k, double* xyz) {
  ARGCHK(f && xyz);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  if (c->lazy) {
    int rc = which == 3 ? frontier_prev_ready(f) : frontier_cells_ready(f);
    if (rc) return rc;
  }
  if (!f->map) {
    fuelmi_set_error("fuelmi_frontier_cluster_centres: the map of this finder has been destroyed");
    return FUELMI_EINVAL;
  }
  {
    const int rcf = frontier_fetch_cluster(f, c);
    if (rcf) return rcf;
  }
  const Geo& g = f->map->g;
  const size_t n = c->size();
  // (a cluster started by an NQ seed keeps the seed apart from its sorted list: materialise the merged order first)
  std::vector<int> merged;
  const int* adr = nullptr;
  if (c->lazy && c->lazy_seed < 0)
    adr = c->lazy;
  else if (!c->lazy)
    adr = c->cells.data();
  else {
    merged.resize(n);
    c->copy_to(merged.data());
    adr = merged.data();
  }
  const double res = g.res, ox = g.org[0], oy = g.org[1], oz = g.org[2];
  const unsigned nz = (unsigned)g.nz, nyz = (unsigned)g.nyz;
  auto decode = [&](size_t i0, size_t i1) {
    // address of z = 0 of the current z-line; the start value makes the first cell decode whatever its address is
    // (addresses are below 2^31; 0xFFFFFFFF made a - lb wrap to a + 1 < nz for cells of the column x = y = 0)
    unsigned lb = 0x80000000u;
    double cx = 0.0, cy = 0.0;
    for (size_t i = i0; i < i1; ++i) {
      const unsigned a = (unsigned)adr[i];
      unsigned z = a - lb;
      if (z >= nz) {  // another line
        const unsigned x = a / nyz, r = a - x * nyz, y = r / nz;
        z = r - y * nz;
        lb = a - z;
        cx = (x + 0.5) * res + ox, cy = (y + 0.5) * res + oy;
      }
      xyz[3 * i] = cx, xyz[3 * i + 1] = cy, xyz[3 * i + 2] = (z + 0.5) * res + oz;
    }
  };
  // a map-spanning cluster (140 k cells = 3.4 MB of doubles) is decoded by the caller and three helper threads of the
  // library: one host thread writing it was the longest single step of a full-box cycle through the facade
  if (n >= 32768)
    host_parallel_for(n, decode);
  else
    decode(0, n);
  return FUELMI_OK;
}
extern "C" int fuelmi_frontier_cluster_filtered_size(const fuelmi_frontier* f, int which, int k) {
  ARGCHK(f);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  return (int)(c->filtered.size() / 3);
}
extern "C" int fuelmi_frontier_cluster_filtered(const fuelmi_frontier* f, int which, int k, float* xyz) {
  ARGCHK(f && xyz);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  if (!c->filtered.empty()) memcpy(xyz, c->filtered.data(), c->filtered.size() * sizeof(float));
  return FUELMI_OK;
}
extern "C" int fuelmi_frontier_cluster_info(const fuelmi_frontier* f, int which, int k, double out9[9]) {
  ARGCHK(f && out9);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  for (int i = 0; i < 3; ++i) out9[i] = c->avg[i], out9[3 + i] = c->bmin[i], out9[6 + i] = c->bmax[i];
  return FUELMI_OK;
}
extern "C" int fuelmi_frontier_removed_count(const fuelmi_frontier* f) {
  ARGCHK(f);
  return (int)f->removed_ids.size();
}
extern "C" int fuelmi_frontier_removed_ids(const fuelmi_frontier* f, int* ids) {
  ARGCHK(f && ids);
  memcpy(ids, f->removed_ids.data(), f->removed_ids.size() * sizeof(int));
  return FUELMI_OK;
}
extern "C" int fuelmi_frontier_get_flags(fuelmi_frontier* f, char* flags) {
  ARGCHK(f && flags);
  FRONTIER_HAS_MAP(f);
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  long n = m->g.N;
  int rc = frontier_ensure_stage(f, (size_t)n);
  if (rc) return rc;
  {
    const int rcr = frontier_apply_reset(f);
    if (rcr) return rcr;
  }
  k_expand_flag_bits<<<fblocks(n, 256), 256, 0, f->stream>>>(f->flag.p, n, (char*)f->d_stage);
  FDBG("k_expand_flag_bits");
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(flags, f->d_stage, (size_t)n, hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return FUELMI_OK;
}


// ---- measurement driver (bench.py): the plan cycle issued from C++ ---------------------------------------
extern "C" int fuelmi_bench_cycles(fuelmi_map* m, fuelmi_frontier* f, fuelmi_bspline_dev* batch, const double ub_min[3],
                                   const double ub_max[3], int n, int serial, int* n_clusters, double* seconds) {
  ARGCHK(m && f && ub_min && ub_max && n >= 0 && n_clusters && seconds && f->map == m);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(frontier_drain(f));
  int rc = FUELMI_OK, ncl = 0;
  f->wait_us_acc = 0.0;
  using clk = std::chrono::steady_clock;
  // host time of every C-ABI call of the cycle (seven clock reads per cycle, ~0.2 us): fuelmi_bench_host_profile
  double hp[6] = {0, 0, 0, 0, 0, 0};
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  const auto t0 = clk::now();
  for (int k = 0; k < n && rc == FUELMI_OK; ++k) {
    const auto a0 = clk::now();
    if ((rc = fuelmi_frontier_reset(f))) break;
    if ((rc = fuelmi_map_set_updated_box(m, ub_min, ub_max))) break;
    const auto a1 = clk::now();
    if (!serial && (rc = fuelmi_frontier_search_begin(f))) break;
    const auto a2 = clk::now();
    if ((rc = fuelmi_map_inflate_local(m))) break;
    const auto a3 = clk::now();
    if ((rc = fuelmi_map_update_esdf(m))) break;
    const auto a4 = clk::now();
    if (batch && (rc = fuelmi_bspline_dev_eval(batch))) break;
    const auto a5 = clk::now();
    if (serial) {
      HIPCHK(hipStreamSynchronize(m->stream));
      if ((rc = fuelmi_frontier_search_begin(f))) break;
    }
    if ((rc = fuelmi_frontier_search_end(f, &ncl))) break;
    const auto a6 = clk::now();
    hp[0] += us(a0, a1), hp[1] += us(a1, a2), hp[2] += us(a2, a3), hp[3] += us(a3, a4), hp[4] += us(a4, a5), hp[5] += us(a5, a6);
  }
  if (rc) return rc;
  HIPCHK(stream_wait(m->stream));
  HIPCHK(frontier_drain(f));
  f->tail_pending = false;
  *seconds = std::chrono::duration<double>(clk::now() - t0).count();
  *n_clusters = ncl;
  for (int q = 0; q < 6; ++q) m->bench_host_us[q] = n ? hp[q] / n : 0.0;
  m->bench_host_us[6] = f->wait_us_acc / std::max(n, 1);
  f->wait_us_acc = 0.0;
  g_ht.report();
  return FUELMI_OK;
}
extern "C" int fuelmi_bench_host_profile(const fuelmi_map* m, double out7[7]) {
  ARGCHK(m && out7);
  for (int q = 0; q < 7; ++q) out7[q] = m->bench_host_us[q];
  return FUELMI_OK;
}

// The same cycle with its results DELIVERED to the host containers the reference's callers read
// (fast_exploration_manager.cpp:99-114: the cluster cell lists of searchFrontiers; planner_manager.cpp:296-314: the
// cost and gradient of every candidate): after every search the cells of all new clusters are copied into
// cells_out (cluster after cluster, as many as fit), after every evaluation cost[C] / grad[C * nvar] are downloaded.
// seconds3: [0] elapsed wall time, [1] of it in the cell copies, [2] in the cost / gradient download.
extern "C" int fuelmi_bench_cycles_delivered(fuelmi_map* m, fuelmi_frontier* f, fuelmi_bspline_dev* batch,
                                             const double ub_min[3], const double ub_max[3], int n, int* cells_out,
                                             size_t cells_cap, double* cost, double* grad, int* n_clusters,
                                             double* seconds3) {
  ARGCHK(m && f && ub_min && ub_max && n >= 0 && n_clusters && seconds3 && cells_out && f->map == m);
  ARGCHK(!batch || (cost && grad));
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(frontier_drain(f));
  int rc = FUELMI_OK, ncl = 0;
  double t_cells = 0.0, t_cg = 0.0;
  using clk = std::chrono::steady_clock;
  const bool kept = f->keep_prev;
  (void)fuelmi_frontier_keep_previous(f, 1);
  // Results are consumed one cycle behind the device: while cycle k runs, the host copies out what cycle k - 1 found
  // -- its cluster cells from the retired buffer set (list 3), its costs and gradients from the pinned slot the
  // B-spline kernel wrote them to.  Every cycle's results are delivered (the last one's after the loop); nothing
  // blocks on a copy engine.
  auto deliver = [&](int cyc, int ncl_of) -> int {
    const auto ta = clk::now();
    size_t at = 0;
    int r2 = FUELMI_OK;
    const int which = cyc < 0 ? 0 : 3;  // (the last cycle has not been retired: its clusters are still list 0)
    const int cnt = fuelmi_frontier_count(f, which);
    for (int c = 0; c < cnt && c < ncl_of && r2 == FUELMI_OK; ++c) {
      const int sz = fuelmi_frontier_cluster_size(f, which, c);
      if (sz < 0 || at + (size_t)sz > cells_cap) break;
      r2 = fuelmi_frontier_cluster_cells(f, which, c, cells_out + at);
      at += (size_t)sz;
    }
    const auto tb = clk::now();
    if (r2 == FUELMI_OK && batch) r2 = fuelmi_bspline_dev_collect(batch, (cyc < 0 ? n - 1 : cyc) & 1, cost, grad);
    const auto tc = clk::now();
    t_cells += std::chrono::duration<double>(tb - ta).count();
    t_cg += std::chrono::duration<double>(tc - tb).count();
    return r2;
  };
  const auto t0 = clk::now();
  int ncl_prev = 0;
  for (int k = 0; k < n && rc == FUELMI_OK; ++k) {
    if ((rc = fuelmi_frontier_reset(f))) break;  // (retires cycle k - 1's clusters to list 3)
    if ((rc = fuelmi_map_set_updated_box(m, ub_min, ub_max))) break;
    if ((rc = fuelmi_frontier_search_begin(f))) break;
    if ((rc = fuelmi_map_inflate_local(m))) break;
    if ((rc = fuelmi_map_update_esdf(m))) break;
    if (batch && (rc = fuelmi_bspline_dev_eval_pinned(batch, k & 1))) break;
    if (k > 0 && (rc = deliver(k - 1, ncl_prev))) break;
    if ((rc = fuelmi_frontier_search_end(f, &ncl))) break;
    ncl_prev = ncl;
  }
  if (rc == FUELMI_OK && n > 0) rc = deliver(-1, ncl_prev);
  (void)fuelmi_frontier_keep_previous(f, kept ? 1 : 0);
  if (rc) return rc;
  HIPCHK(stream_wait(m->stream));
  HIPCHK(frontier_drain(f));
  f->tail_pending = false;
  seconds3[0] = std::chrono::duration<double>(clk::now() - t0).count();
  seconds3[1] = t_cells;
  seconds3[2] = t_cg;
  *n_clusters = ncl;
  return FUELMI_OK;
}

// Measurement driver for the streaming cycle (one depth frame per cycle), issued from C++ like fuelmi_bench_cycles.
extern "C" int fuelmi_bench_stream(fuelmi_map* m, fuelmi_frontier* f, fuelmi_bspline_dev* batch, int n,
                                   const void* const* depth, int rows, int cols, const fuelmi_depth_cfg* cfg,
                                   const double* cam_pos3, const double* cam_q4, int serial, int* n_clusters,
                                   double* box_voxels, double* seconds) {
  ARGCHK(m && f && n >= 0 && depth && cfg && cam_pos3 && cam_q4 && n_clusters && seconds && f->map == m);
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(frontier_drain(f));
  int rc = FUELMI_OK, ncl = 0;
  double vox = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  // Frame k + 1 is fused while the search of frame k is still running: the search reads the occupancy planes only in
  // its first kernels, the fusion waits for those ON THE DEVICE (map_wait_plane_readers), and the host's wait for the
  // fused frame's box (which the next search needs) falls beside the chain instead of in front of it.  Per frame: the
  // search bookkeeping (collect + commit the previous search, begin this one), the map chain (inflation, ESDF, B-spline
  // batch), the next fusion.  The frame is bound by the host's ~55 us of API calls plus the device's fusion ->
  // plane-reading kernels -> fusion chain (FUELMI_STREAM_TIMING=1: host time per call group, =2: also a device timeline
  // from events); issuing the map chain BEFORE the bookkeeping (measured in round 4) starts the map stream ~25 us
  // earlier and the search chain as much later: 1-2 % slower.  Same calls, same arguments, same results as the
  // frame-by-frame order (serial != 0 keeps that order for diagnostics).
  int npts = 0;
  if (n > 0)
    rc = fuelmi_map_input_depth(m, static_cast<const unsigned short*>(depth[0]), rows, cols, cfg, cam_pos3, cam_q4, &npts);
  auto map_chain = [&]() -> int {
    int r = FUELMI_OK;
    if (npts > 0) {
      int lo[3], hi[3];
      if ((r = fuelmi_map_get_local_bound(m, lo, hi))) return r;
      vox += (double)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
      if ((r = fuelmi_map_inflate_local(m))) return r;
      if ((r = fuelmi_map_update_esdf(m))) return r;
    }
    if (batch) r = fuelmi_bspline_dev_eval(batch);
    return r;
  };
  auto fuse_next = [&](int k, int* np) -> int {
    return fuelmi_map_input_depth(m, static_cast<const unsigned short*>(depth[k + 1]), rows, cols, cfg, cam_pos3 + 3 * (k + 1),
                                  cam_q4 + 4 * (k + 1), np);
  };
  if (serial) {
    for (int k = 0; k < n && rc == FUELMI_OK; ++k) {
      if ((rc = map_chain())) break;
      int npts_next = 0;
      HIPCHK(hipStreamSynchronize(m->stream));
      if ((rc = fuelmi_frontier_search_begin(f))) break;
      if ((rc = fuelmi_frontier_search_end(f, &ncl))) break;
      if ((rc = fuelmi_frontier_commit(f, 0))) break;
      if (k + 1 < n && (rc = fuse_next(k, &npts_next))) break;
      npts = npts_next;
    }
  } else {
    static const bool timing = getenv("FUELMI_STREAM_TIMING") != nullptr;  // host wall clock of every call of the loop
    double acc[5] = {0, 0, 0, 0, 0};
    auto tick = [&](int slot, std::chrono::steady_clock::time_point& t) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      acc[slot] += std::chrono::duration<double, std::micro>(now - t).count();
      t = now;
    };
    static const bool timeline = timing && atoi(getenv("FUELMI_STREAM_TIMING")) >= 2;  // + device timeline (events)
    std::vector<hipEvent_t> tl;  // per frame: chain start, planes read, resolved, tail done, fusion done, map chain done
    if (timeline) {
      tl.resize((size_t)n * 6);
      for (auto& e : tl) HIPCHK(hipEventCreate(&e));
    }
    // (Round 4 also tried a second host thread for the search bookkeeping, like the reference's separate map and
    // planning callbacks: launches from two threads serialise inside the HIP runtime and each gets slower -- 7.4 k
    // frames/s against 8.1 k.  One thread issues everything.)
    auto bookkeeping = [&](int k) -> int {
      int r = FUELMI_OK;
      if (k > 0) {
        if ((r = fuelmi_frontier_search_end(f, &ncl))) return r;
        if ((r = fuelmi_frontier_commit(f, 0))) return r;
      }
      return fuelmi_frontier_search_begin(f);
    };
    for (int k = 0; k < n && rc == FUELMI_OK; ++k) {
      auto t = std::chrono::steady_clock::now();
      if (timeline) {
        if (hipEventRecord(tl[(size_t)k * 6 + 4], m->stream) != hipSuccess) rc = FUELMI_EHIP;  // (the fusion of this frame is queued)
        f->tl_ev = &tl[(size_t)k * 6];
      }
      if (rc == FUELMI_OK) rc = bookkeeping(k);
      tick(3, t);
      if (rc == FUELMI_OK) rc = map_chain();
      if (timeline && rc == FUELMI_OK && hipEventRecord(tl[(size_t)k * 6 + 5], m->stream) != hipSuccess) rc = FUELMI_EHIP;
      tick(0, t);
      if (rc) break;
      int npts_next = 0;
      if (k + 1 < n && (rc = fuse_next(k, &npts_next))) break;
      tick(4, t);
      npts = npts_next;
    }
    if (timeline) {
      f->tl_ev = nullptr;
      HIPCHK(hipStreamSynchronize(m->stream));
      HIPCHK(frontier_drain(f));
      double off[6] = {0, 0, 0, 0, 0, 0}, period = 0;
      int cnt = 0;
      for (int k = 10; k + 1 < n; ++k) {
        float ms = 0.f;
        const hipEvent_t base = tl[(size_t)k * 6 + 4];  // fusion of frame k done
        bool okf = true;
        double o[6];
        for (int j = 0; j < 6 && okf; ++j) {
          okf = hipEventElapsedTime(&ms, base, tl[(size_t)k * 6 + j]) == hipSuccess;
          o[j] = ms * 1e3;
        }
        if (okf) okf = hipEventElapsedTime(&ms, base, tl[(size_t)(k + 1) * 6 + 4]) == hipSuccess;
        if (!okf) {
          (void)hipGetLastError();
          continue;
        }
        for (int j = 0; j < 6; ++j) off[j] += o[j];
        period += ms * 1e3;
        ++cnt;
      }
      if (cnt)
        std::fprintf(stderr, "[stream-timing] device timeline, us after the frame's fusion finished: map chain done %.1f; search "
                     "chain starts %.1f, planes read %.1f, resolved %.1f, tail done %.1f; next frame's fusion done %.1f (%d frames)\n",
                     off[5] / cnt, off[0] / cnt, off[1] / cnt, off[2] / cnt, off[3] / cnt, period / cnt, cnt);
      for (auto& e : tl) (void)hipEventDestroy(e);
    }
    if (timing && n > 0)
      std::fprintf(stderr, "[stream-timing] host us per frame: map chain %.1f, search bookkeeping (collect, commit, begin) %.1f, "
                   "input_depth (blocked on the device) %.1f\n", acc[0] / n, acc[3] / n, acc[4] / n);
    if (rc == FUELMI_OK && n > 0) {
      if ((rc = fuelmi_frontier_search_end(f, &ncl)) == FUELMI_OK) rc = fuelmi_frontier_commit(f, 0);
    }
  }
  if (rc) return rc;
  HIPCHK(stream_wait(m->stream));
  HIPCHK(frontier_drain(f));
  f->tail_pending = false;
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *n_clusters = ncl;
  if (box_voxels) *box_voxels = vox;
  return FUELMI_OK;
}
