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
#include <cmath>
#include <cstring>
#include <list>
#include <unordered_map>
#include <vector>

#include "fuelmi_internal.h"

#define NOCLAIM 0xFFFFFFFFu

struct FArgs {
  Box3 qbox;  // Q0 index box (isInBox & z >= iz_min), inclusive
  Box3 sbox;  // scanned index box, inclusive
  int w0;     // first word processed (multiple of 256)
  int nwords; // words processed (multiple of 256)
  u32 cap_q, cap_s;
  int cluster_min;
  const u64* occ;
  const u64* unk;
  u64* flag;
  u64* qb;
  u64* sb;
  u64* pref;       // per word: packed in-block exclusive prefix (lo = q count, hi = s count)
  u64* blocksum;   // per 256-word block: packed totals
  u64* blockscan;  // exclusive scan of blocksum
  u32* counts;     // [0]=nq [1]=ns [2]=overflow [3]=n_kept
  u32* cell_adr;   // [cap_q]
  u32* parent;     // [cap_q]
  u32* claim;      // [cap_q] per root: claimer address
  int* cell_slot;  // [cap_q] slot of the owning cluster if kept, else -1
  u32* seed_adr;   // [cap_s]
  u32* csize;      // [cap_q + cap_s] cluster sizes by slot
  u32* kept;       // [.. x 3] (claimer address, slot, size)
  u32 cap_kept;
};

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

// ---- kernels ----------------------------------------------------------------------------------
// isFrontierChanged (:365-372) for the cells of several clusters: changed[cl] |= !F1(cell)
__global__ void k_check_clusters(Geo g, const u64* __restrict__ occ, const u64* __restrict__ unk,
                                 const int* __restrict__ cells, const int* __restrict__ cell_cluster, int n,
                                 int* __restrict__ changed) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!f1_cell(g, occ, unk, cells[i])) changed[cell_cluster[i]] = 1;
}
__global__ void k_clear_flags(u64* flag, const int* __restrict__ cells, const int* __restrict__ cell_cluster,
                              const int* __restrict__ changed, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (changed[cell_cluster[i]]) {
    long a = cells[i];
    atomicAnd(&flag[a >> 6], ~(1ull << (a & 63)));
  }
}

// predicate planes + in-block packed prefix of their popcounts
__global__ void __launch_bounds__(256) k_pred(Geo g, FArgs F) {
  __shared__ u64 wsum[4];
  const int rel = blockIdx.x * 256 + threadIdx.x;
  const int w = F.w0 + rel;
  u64 q = 0ull, s = 0ull;
  if (w < g.W) {
    u64 z0, zl, y0, yl, mq, ms;
    word_masks(g, w, F.qbox, F.sbox, z0, zl, y0, yl, mq, ms);
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

// exclusive scan of the block sums (single block) + totals
__global__ void __launch_bounds__(256) k_scan_sums(FArgs F, int nblocks) {
  __shared__ u64 part[256];
  const int per = (nblocks + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
  u64 s = 0;
  for (int b = b0; b < b1; ++b) s += F.blocksum[b];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 run = 0;
    for (int t = 0; t < 256; ++t) {
      u64 v = part[t];
      part[t] = run;
      run += v;
    }
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
  }
  __syncthreads();
  u64 run = part[threadIdx.x];
  for (int b = b0; b < b1; ++b) {
    u64 v = F.blocksum[b];
    F.blockscan[b] = run;
    run += v;
  }
}

__device__ __forceinline__ u32 rank_q(const FArgs& F, long a) {
  int w = (int)(a >> 6);
  int rel = w - F.w0;
  u64 pk = F.blockscan[rel >> 8] + F.pref[rel];
  return (u32)pk + (u32)__popcll(F.qb[w] & ((1ull << (a & 63)) - 1ull));
}
__device__ __forceinline__ u32 rank_s(const FArgs& F, long a) {
  int w = (int)(a >> 6);
  int rel = w - F.w0;
  u64 pk = F.blockscan[rel >> 8] + F.pref[rel];
  return (u32)(pk >> 32) + (u32)__popcll(F.sb[w] & ((1ull << (a & 63)) - 1ull));
}

// ordered compaction of Q0 cells and NQ seeds
__global__ void __launch_bounds__(256) k_compact(Geo g, FArgs F) {
  const int rel = blockIdx.x * 256 + threadIdx.x;
  const int w = F.w0 + rel;
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

__device__ __forceinline__ u32 uf_find(const u32* parent, u32 i) {
  u32 p = parent[i];
  while (p != i) {
    i = p;
    p = parent[i];
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

// 26-connectivity: union with the 13 neighbours of lower address
__global__ void __launch_bounds__(256) k_union(Geo g, FArgs F) {
  const u32 nq = F.counts[0];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
    long a = F.cell_adr[i];
    int x = (int)(a / g.nyz);
    int r = (int)(a - (long)x * g.nyz);
    int y = r / g.nz, z = r - y * g.nz;
    for (int k = 0; k < 13; ++k) {
      int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;  // k=0..12: lexicographically < 0
      int xx = x + dx, yy = y + dy, zz = z + dz;
      if (xx < 0 || yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz) continue;
      long an = a + (long)dx * g.nyz + (long)dy * g.nz + dz;
      if (!((F.qb[an >> 6] >> (an & 63)) & 1ull)) continue;
      u32 j = rank_q(F, an);
      if (j < F.cap_q) uf_union(F.parent, i, j);
    }
  }
}

__global__ void __launch_bounds__(256) k_flatten(Geo g, FArgs F) {
  const u32 nq = F.counts[0];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x)
    F.parent[i] = uf_find(F.parent, i);
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
  const u32 nq_r = (nq + 63u) & ~63u, ns_r = (ns + 63u) & ~63u;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nq_r; i += gridDim.x * blockDim.x) {
    bool active = false;
    u32 a = 0, r = 0;
    if (i < nq) {
      a = F.cell_adr[i];
      if (in_box(g, F.sbox, a)) {
        active = true;
        r = F.parent[i];
      }
    }
    wave_min_claim(F.claim, active, r, a);
  }
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < ns_r; i += gridDim.x * blockDim.x) {
    const bool live = i < ns;
    long a = live ? F.seed_adr[i] : 0;
    int x = (int)(a / g.nyz);
    int rr = (int)(a - (long)x * g.nyz);
    int y = rr / g.nz, z = rr - y * g.nz;
    u32 last = NOCLAIM;
    for (int k = 0; k < 27; ++k) {
      if (k == 13) continue;
      int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
      int xx = x + dx, yy = y + dy, zz = z + dz;
      bool active = live && !(xx < 0 || xx >= g.nx || yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz);
      u32 r = 0;
      if (active) {
        long an = a + (long)dx * g.nyz + (long)dy * g.nz + dz;
        active = (F.qb[an >> 6] >> (an & 63)) & 1ull;
        if (active) {
          u32 j = rank_q(F, an);
          active = j < F.cap_q;
          if (active) {
            r = F.parent[j];
            if (r == last) active = false;  // this seed already claimed that component
            last = r;
          }
        }
      }
      if (__ballot(active)) wave_min_claim(F.claim, active, r, (u32)a);
    }
  }
}

// cluster sizes by slot: own claimer -> its compact index; NQ seed claimer -> nq + seed rank.
// lanes of a wave that hit the same slot are merged into one atomic.
__global__ void __launch_bounds__(256) k_sizes(Geo g, FArgs F) {
  const u32 nq = F.counts[0];
  const u32 total = (nq + 63u) & ~63u;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    bool active = false;
    u32 slot = 0;
    if (i < nq) {
      u32 cl = F.claim[F.parent[i]];
      if (cl != NOCLAIM) {
        active = true;
        bool own = (F.qb[cl >> 6] >> (cl & 63)) & 1ull;
        slot = own ? rank_q(F, cl) : nq + rank_s(F, cl);
        F.cell_slot[i] = (int)slot;
      } else
        F.cell_slot[i] = -1;
    }
    u64 todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
      int leader = __builtin_ctzll(todo);
      u32 k = __shfl(slot, leader, 64);
      u64 same = __ballot(active && slot == k) & todo;
      if (lane == leader) atomicAdd(&F.csize[k], (u32)__popcll(same));
      todo &= ~same;
    }
  }
}

// flags (all claimed cells + all NQ seeds), kept-cluster list, per-cell kept slot
__global__ void __launch_bounds__(256) k_finalize(Geo g, FArgs F) {
  const int rel = blockIdx.x * 256 + threadIdx.x;
  const int w = F.w0 + rel;
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

__global__ void k_expand_flag_bits(const u64* __restrict__ bits, long n, char* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (char)((bits[i >> 6] >> (i & 63)) & 1ull);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct HCluster {
  std::vector<int> cells;  // ascending voxel addresses
  double avg[3], bmin[3], bmax[3];
};

struct fuelmi_frontier {
  fuelmi_map* map = nullptr;
  fuelmi_frontier_cfg cfg;
  int iz_min = 0;
  Plane flag, qb, sb;
  FArgs F;
  size_t nwords_alloc = 0;
  std::vector<void*> allocs;
  std::list<HCluster> frontiers, dormant, tmp;
  std::vector<int> removed_ids;
  void* h_pin = nullptr;  // pinned result staging
  size_t pin_bytes = 0;
  std::vector<int> slot2rank;
};

static inline int fblocks(long n, int t, int cap = 1 << 16) {
  long b = (n + t - 1) / t;
  return (int)std::max(1L, std::min((long)cap, b));
}

static void cluster_info(const fuelmi_map* m, HCluster& c) {
  // computeFrontierInfo (:374-390): mean and AABB of the voxel centres
  const Geo& g = m->g;
  for (int k = 0; k < 3; ++k) c.avg[k] = 0.0;
  bool first = true;
  for (int a : c.cells) {
    int x = a / g.nyz, r = a - x * g.nyz, y = r / g.nz, z = r - y * g.nz;
    const int id[3] = {x, y, z};
    for (int k = 0; k < 3; ++k) {
      double p = (id[k] + 0.5) * g.res + g.org[k];
      c.avg[k] += p;
      if (first) {
        c.bmin[k] = c.bmax[k] = p;
      } else {
        c.bmin[k] = std::min(c.bmin[k], p);
        c.bmax[k] = std::max(c.bmax[k], p);
      }
    }
    first = false;
  }
  for (int k = 0; k < 3; ++k) c.avg[k] /= double(c.cells.size());
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
  return FUELMI_OK;
}

extern "C" void fuelmi_frontier_destroy(fuelmi_frontier* f) {
  if (!f) return;
  (void)hipSetDevice(f->map->device);
  (void)hipStreamSynchronize(f->map->stream);
  for (void* p : f->allocs) (void)hipFree(p);
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  Plane* pl[] = {&f->flag, &f->qb, &f->sb};
  for (Plane* p : pl)
    if (p->base) (void)hipFree(p->base);
  delete f;
}

extern "C" int fuelmi_frontier_create(fuelmi_map* m, const fuelmi_frontier_cfg* cfg, fuelmi_frontier** out) {
  ARGCHK(m && cfg && out);
  *out = nullptr;
  HIPCHK(hipSetDevice(m->device));
  fuelmi_frontier* f = new fuelmi_frontier;
  f->map = m;
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
  if ((rc = plane_alloc(m, f->flag)) || (rc = plane_alloc(m, f->qb)) || (rc = plane_alloc(m, f->sb))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  FArgs& F = f->F;
  memset(&F, 0, sizeof(F));
  size_t nwords = ((size_t)g.W + 255) / 256 * 256 + 256;
  f->nwords_alloc = nwords;
  F.cap_q = (u32)std::max<size_t>(1u << 20, (size_t)g.N / 8);
  F.cap_s = (u32)std::max<size_t>(1u << 18, (size_t)g.N / 32);
  F.cap_kept = 1u << 16;
  F.cluster_min = cfg->cluster_min;
  if ((rc = dmalloc(f, &F.pref, nwords)) || (rc = dmalloc(f, &F.blocksum, nwords / 256 + 1)) ||
      (rc = dmalloc(f, &F.blockscan, nwords / 256 + 1)) || (rc = dmalloc(f, &F.counts, 8)) ||
      (rc = dmalloc(f, &F.cell_adr, F.cap_q)) || (rc = dmalloc(f, &F.parent, F.cap_q)) ||
      (rc = dmalloc(f, &F.claim, F.cap_q)) || (rc = dmalloc(f, &F.cell_slot, F.cap_q)) ||
      (rc = dmalloc(f, &F.seed_adr, F.cap_s)) || (rc = dmalloc(f, &F.csize, (size_t)F.cap_q + F.cap_s)) ||
      (rc = dmalloc(f, &F.kept, (size_t)F.cap_kept * 3))) {
    fuelmi_frontier_destroy(f);
    return rc;
  }
  F.occ = m->occ_bits.p;
  F.unk = m->unk_bits.p;
  F.flag = f->flag.p;
  F.qb = f->qb.p;
  F.sb = f->sb.p;
  HIPCHK(hipStreamSynchronize(m->stream));
  *out = f;
  return FUELMI_OK;
}

// drop clusters of `L` that overlap the updated box and contain a cell that is no longer a
// frontier (searchFrontiers :62-93); flags of dropped clusters are cleared on the device
static int remove_changed(fuelmi_frontier* f, std::list<HCluster>& L, const double* umin, const double* umax,
                          std::vector<int>* removed_ids) {
  fuelmi_map* m = f->map;
  std::vector<std::list<HCluster>::iterator> cand;
  std::vector<int> cand_pos;
  int pos = 0;
  size_t ncell = 0;
  for (auto it = L.begin(); it != L.end(); ++it, ++pos)
    if (have_overlap(it->bmin, it->bmax, umin, umax)) {
      cand.push_back(it);
      cand_pos.push_back(pos);
      ncell += it->cells.size();
    }
  if (cand.empty()) return FUELMI_OK;
  std::vector<int> cells, cl;
  cells.reserve(ncell);
  cl.reserve(ncell);
  for (size_t k = 0; k < cand.size(); ++k)
    for (int a : cand[k]->cells) {
      cells.push_back(a);
      cl.push_back((int)k);
    }
  size_t bytes = ncell * sizeof(int);
  int rc = map_ensure_stage(m, 2 * bytes + cand.size() * sizeof(int) + 64, 0);
  if (rc) return rc;
  int* d_cells = (int*)m->d_stage;
  int* d_cl = d_cells + ncell;
  int* d_changed = d_cl + ncell;
  HIPCHK(hipMemcpyAsync(d_cells, cells.data(), bytes, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(d_cl, cl.data(), bytes, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemsetAsync(d_changed, 0, cand.size() * sizeof(int), m->stream));
  k_check_clusters<<<fblocks((long)ncell, 256), 256, 0, m->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, d_cells,
                                                                    d_cl, (int)ncell, d_changed);
  k_clear_flags<<<fblocks((long)ncell, 256), 256, 0, m->stream>>>(f->flag.p, d_cells, d_cl, d_changed, (int)ncell);
  std::vector<int> changed(cand.size());
  HIPCHK(hipMemcpyAsync(changed.data(), d_changed, cand.size() * sizeof(int), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  // removed_ids_ semantics (:74-85): index in the list as it shrinks
  int erased = 0;
  for (size_t k = 0; k < cand.size(); ++k)
    if (changed[k]) {
      if (removed_ids) removed_ids->push_back(cand_pos[k] - erased);
      L.erase(cand[k]);
      ++erased;
    }
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_search(fuelmi_frontier* f, int* n_new) {
  ARGCHK(f && n_new);
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  FArgs& F = f->F;
  f->tmp.clear();
  double umin[3], umax[3];
  fuelmi_map_get_updated_box(m, umin, umax, 1);

  StageScope sc(m, FUELMI_K_FRONTIER);
  f->removed_ids.clear();
  int rc = remove_changed(f, f->frontiers, umin, umax, &f->removed_ids);
  if (rc) return rc;
  rc = remove_changed(f, f->dormant, umin, umax, nullptr);
  if (rc) return rc;

  // scan box (:95-106): updated box +- (1,1,0.5) clipped to the exploration box, as indices
  const int nv[3] = {g.nx, g.ny, g.nz};
  bool empty = false;
  for (int k = 0; k < 3; ++k) {
    double infl = (k == 2) ? 0.5 : 1.0;
    double smin = std::max(umin[k] - infl, m->cfg.box_min[k]);
    double smax = std::min(umax[k] + infl, m->cfg.box_max[k]);
    int lo = (int)std::floor((smin - g.org[k]) * g.res_inv);
    int hi = (int)std::floor((smax - g.org[k]) * g.res_inv);
    lo = std::max(lo, 0);
    hi = std::min(hi, nv[k] - 1);  // the reference would index out of the map here (UB)
    F.sbox.lo[k] = lo;
    F.sbox.hi[k] = hi;
    if (lo > hi) empty = true;
  }
  // Q box: isInBox(idx) (min <= id < max) and z >= iz_min, inside the map
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
  *n_new = 0;
  if (empty) return FUELMI_OK;

  // words to process: everything the BFS could reach = Q box, plus the scan box
  auto adr = [&](const int* id) { return (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2]; };
  long a_lo = adr(F.sbox.lo), a_hi = adr(F.sbox.hi);
  if (F.qbox.lo[0] <= F.qbox.hi[0] && F.qbox.lo[1] <= F.qbox.hi[1] && F.qbox.lo[2] <= F.qbox.hi[2]) {
    a_lo = std::min(a_lo, adr(F.qbox.lo));
    a_hi = std::max(a_hi, adr(F.qbox.hi));
  }
  F.w0 = (int)((a_lo >> 6) & ~255L);
  int w_hi = (int)(a_hi >> 6);
  int nblocks = (w_hi - F.w0) / 256 + 1;
  F.nwords = nblocks * 256;

  k_pred<<<nblocks, 256, 0, m->stream>>>(g, F);
  k_scan_sums<<<1, 256, 0, m->stream>>>(F, nblocks);
  k_compact<<<nblocks, 256, 0, m->stream>>>(g, F);
  const int cgrid = 2048;
  k_union<<<cgrid, 256, 0, m->stream>>>(g, F);
  k_flatten<<<cgrid, 256, 0, m->stream>>>(g, F);
  k_claim<<<cgrid, 256, 0, m->stream>>>(g, F);
  k_sizes<<<cgrid, 256, 0, m->stream>>>(g, F);
  k_finalize<<<nblocks, 256, 0, m->stream>>>(g, F);
  HIPCHK(hipGetLastError());

  // results come back through one pinned staging buffer: [counts | kept | cell_adr | cell_slot]
  if (!f->h_pin) {
    f->pin_bytes = 64 + (size_t)F.cap_kept * 12 + (size_t)F.cap_q * 8;
    HIPCHK(hipHostMalloc(&f->h_pin, f->pin_bytes, hipHostMallocDefault));
    f->slot2rank.assign((size_t)F.cap_q + F.cap_s, -1);
  }
  u32* counts = reinterpret_cast<u32*>(f->h_pin);
  HIPCHK(hipMemcpyAsync(counts, F.counts, 4 * sizeof(u32), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  if (counts[2] || counts[3] > F.cap_kept) {
    fuelmi_set_error("frontier capacity exceeded (cells %u/%u seeds %u/%u clusters %u/%u)", counts[0], F.cap_q,
                     counts[1], F.cap_s, counts[3], F.cap_kept);
    return FUELMI_ELIMIT;
  }
  const u32 nq = counts[0], nkept = counts[3];
  if (nkept == 0) return FUELMI_OK;
  u32* h_kept = counts + 16;
  u32* h_adr = h_kept + (size_t)F.cap_kept * 3;
  int* h_slot = reinterpret_cast<int*>(h_adr + F.cap_q);
  HIPCHK(hipMemcpyAsync(h_kept, F.kept, (size_t)nkept * 3 * sizeof(u32), hipMemcpyDeviceToHost, m->stream));
  if (nq) {
    HIPCHK(hipMemcpyAsync(h_adr, F.cell_adr, (size_t)nq * sizeof(u32), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(h_slot, F.cell_slot, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, m->stream));
  }
  HIPCHK(hipStreamSynchronize(m->stream));

  // assemble: clusters in creation order (= ascending claimer address); cells arrive in ascending
  // address order, so appending keeps every cluster sorted -- only an NQ seed needs inserting
  std::vector<u32> order(nkept);
  for (u32 k = 0; k < nkept; ++k) order[k] = k;
  std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return h_kept[3 * a] < h_kept[3 * b]; });
  std::vector<HCluster*> by_rank(nkept);
  for (u32 r = 0; r < nkept; ++r) {
    const u32 k = order[r];
    f->tmp.emplace_back();
    HCluster& c = f->tmp.back();
    c.cells.reserve(h_kept[3 * k + 2]);
    f->slot2rank[h_kept[3 * k + 1]] = (int)r;
    by_rank[r] = &c;
  }
  for (u32 i = 0; i < nq; ++i) {
    int s = h_slot[i];
    if (s < 0) continue;
    int r = f->slot2rank[s];
    if (r >= 0) by_rank[r]->cells.push_back((int)h_adr[i]);
  }
  for (u32 r = 0; r < nkept; ++r) {
    const u32 k = order[r];
    const u32 slot = h_kept[3 * k + 1];
    f->slot2rank[slot] = -1;
    HCluster& c = *by_rank[r];
    if (slot >= nq) {  // the NQ seed belongs to its cluster
      int a = (int)h_kept[3 * k];
      c.cells.insert(std::lower_bound(c.cells.begin(), c.cells.end(), a), a);
    }
    cluster_info(m, c);
  }
  *n_new = (int)f->tmp.size();
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_reset(fuelmi_frontier* f) {
  ARGCHK(f);
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  f->frontiers.clear();
  f->dormant.clear();
  f->tmp.clear();
  f->removed_ids.clear();
  HIPCHK(hipMemsetAsync(f->flag.p, 0, (size_t)m->g.W * sizeof(u64), m->stream));
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_commit(fuelmi_frontier* f, int dormant) {
  ARGCHK(f);
  auto& dst = dormant ? f->dormant : f->frontiers;
  dst.splice(dst.end(), f->tmp);
  return FUELMI_OK;
}

static const std::list<HCluster>* pick(const fuelmi_frontier* f, int which) {
  return which == 0 ? &f->tmp : (which == 1 ? &f->frontiers : (which == 2 ? &f->dormant : nullptr));
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
  return (int)c->cells.size();
}
extern "C" int fuelmi_frontier_cluster_cells(const fuelmi_frontier* f, int which, int k, int* adr) {
  ARGCHK(f && adr);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  memcpy(adr, c->cells.data(), c->cells.size() * sizeof(int));
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
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  long n = m->g.N;
  int rc = map_ensure_stage(m, (size_t)n, 0);
  if (rc) return rc;
  k_expand_flag_bits<<<fblocks(n, 256), 256, 0, m->stream>>>(f->flag.p, n, (char*)m->d_stage);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(flags, m->d_stage, (size_t)n, hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  return FUELMI_OK;
}
