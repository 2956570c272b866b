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
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <unordered_map>
#include <vector>

#include "fuelmi_internal.h"
#include "frontier_internal.h"

// FUELMI_DEBUG_SYNC=1: synchronise and name every frontier kernel (locates device faults)
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
// global indices [cand_start[k], cand_start[k+1])
__device__ __forceinline__ int pool_cluster_of(const u32* __restrict__ cand_start, int ncand, u32 i) {
  int lo = 0, hi = ncand - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cand_start[mid] <= i)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}
__global__ void k_check_pool(Geo g, const u64* __restrict__ occ, const u64* __restrict__ unk,
                             const u32* __restrict__ pool, const u64* __restrict__ cand_off,
                             const u32* __restrict__ cand_start, int ncand, u32 total, int* __restrict__ changed) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = pool_cluster_of(cand_start, ncand, i);
  const u32 a = pool[cand_off[k] + (i - cand_start[k])];
  if (!f1_cell(g, occ, unk, a)) changed[k] = 1;
}
__global__ void k_clear_pool(u64* flag, const u32* __restrict__ pool, const u64* __restrict__ cand_off,
                             const u32* __restrict__ cand_start, int ncand, u32 total,
                             const int* __restrict__ changed) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = pool_cluster_of(cand_start, ncand, i);
  if (!changed[k]) return;
  const long a = pool[cand_off[k] + (i - cand_start[k])];
  atomicAnd(&flag[a >> 6], ~(1ull << (a & 63)));
}
__global__ void k_pool_put(u32* __restrict__ pool, const u32* __restrict__ cells, const PoolPut* __restrict__ table) {
  const PoolPut e = table[blockIdx.x];  // one workgroup per cluster
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
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

extern "C" void fuelmi_frontier_destroy(fuelmi_frontier* f) {
  if (!f) return;
  (void)hipSetDevice(f->map->device);
  (void)hipStreamSynchronize(f->map->stream);
  if (f->stream) {
    (void)hipStreamSynchronize(f->stream);
    (void)hipStreamDestroy(f->stream);
  }
  if (f->ev_dep) (void)hipEventDestroy(f->ev_dep);
  if (f->d_stage) (void)hipFree(f->d_stage);
  for (void* p : f->allocs) (void)hipFree(p);
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  if (f->h_var) (void)hipHostFree(f->h_var);
  if (f->pool) (void)hipFree(f->pool);
  if (f->h_changed) (void)hipHostFree(f->h_changed);
  frontier_split_free(f);
  frontier_order_free(f);
  for (hipGraphExec_t e : f->graph_exec)
    if (e) (void)hipGraphExecDestroy(e);
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
  F.keys_from_slots = 1;
  F.occ = m->occ_bits.p;
  F.unk = m->unk_bits.p;
  F.flag = f->flag.p;
  F.qb = f->qb.p;
  F.sb = f->sb.p;
  HIPCHK(hipStreamSynchronize(m->stream));
  {
    // the scan is a chain of short latency-bound kernels and is the critical path of a plan cycle:
    // give its stream the highest priority so the wide ESDF kernels of the map stream fill in around it
    int lo_p = 0, hi_p = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    HIPCHK(hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, hi_p));
  }
  HIPCHK(hipEventCreateWithFlags(&f->ev_dep, hipEventDisableTiming));

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
  // results land in one pinned host buffer [counts | cluster records | chunk records | cells]
  f->pin_bytes = 64 + (size_t)F.cap_kept * sizeof(KeptRec) + ((size_t)F.cap_q / SZ_CH + 2) * 40 + (size_t)F.cap_q * 4;
  HIPCHK(hipHostMalloc(&f->h_pin, f->pin_bytes, hipHostMallocDefault));
  F.h_counts = reinterpret_cast<u32*>(f->h_pin);
  F.h_rec = reinterpret_cast<KeptRec*>(F.h_counts + 16);
  F.h_part = reinterpret_cast<u32*>(F.h_rec + F.cap_kept);
  F.h_cells = F.h_part + ((size_t)F.cap_q / SZ_CH + 2) * 10;
  // CCL tile = TX x TY z-lines with u32 labels in LDS (<= 48 KiB so three workgroups share a CU)
  f->TY = 16;
  f->TX = std::max(1, std::min(8, (48 * 1024) / (f->TY * g.nz * 4)));
  if (const char* e = getenv("FUELMI_CCL_TILE")) {  // tuning hook: "TXxTY"
    int a = 0, b = 0;
    if (sscanf(e, "%dx%d", &a, &b) == 2 && a > 0 && b > 0 && (size_t)a * b * g.nz * 4 <= 150 * 1024) f->TX = a, f->TY = b;
  }
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
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_local<256>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->ccl_lds));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_local<512>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->ccl_lds));
    }
  }
  *out = f;
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
  for (std::list<HCluster>* L : {&f->frontiers, &f->dormant})
    for (HCluster& c : *L) live += c.cells.size();
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
  size_t need = 0, nlazy = 0;
  for (HCluster& c : clusters) need += c.size(), nlazy += c.lazy ? 1 : 0;
  int rc = pool_reserve(f, need);
  if (rc) return rc;
  std::vector<PoolPut> table;
  table.reserve(nlazy);
  for (HCluster& c : clusters) {
    if (!c.lazy) {
      if ((rc = pool_upload(f, c))) return rc;
      continue;
    }
    PoolPut e;
    e.dst = f->pool_used;
    e.src = (u32)(c.lazy - reinterpret_cast<const int*>(f->F.h_cells));
    e.n = (u32)c.lazy_n;
    e.seed = c.lazy_seed;
    table.push_back(e);
    c.pool_off = f->pool_used;
    f->pool_used += c.size();
    c.materialize();
  }
  if (table.empty()) return FUELMI_OK;
  if ((rc = frontier_ensure_stage(f, table.size() * sizeof(PoolPut)))) return rc;
  HIPCHK(hipMemcpyAsync(f->d_stage, table.data(), table.size() * sizeof(PoolPut), hipMemcpyHostToDevice, f->stream));
  k_pool_put<<<(unsigned)table.size(), 256, 0, f->stream>>>(f->pool, f->F.ms_val[f->last_fin],
                                                             reinterpret_cast<const PoolPut*>(f->d_stage));
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
    total += (u32)c.cells.size();
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
    f->h_changed = nullptr;
    f->h_changed_cap = 0;
    const size_t cap = nc + nc / 2 + 64;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&f->h_changed), cap * sizeof(int), hipHostMallocDefault));
    f->h_changed_cap = cap;
  }
  const size_t b_off = nc * sizeof(u64), b_start = ((nc * sizeof(u32) + 7) / 8) * 8;
  int rc = frontier_ensure_stage(f, b_off + b_start + nc * sizeof(int) + 64);
  if (rc) return rc;
  u64* d_off = reinterpret_cast<u64*>(f->d_stage);
  u32* d_start = reinterpret_cast<u32*>(reinterpret_cast<unsigned char*>(f->d_stage) + b_off);
  int* d_changed = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(f->d_stage) + b_off + b_start);
  HIPCHK(hipMemcpyAsync(d_off, off.data(), b_off, hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemcpyAsync(d_start, start.data(), nc * sizeof(u32), hipMemcpyHostToDevice, f->stream));
  HIPCHK(hipMemsetAsync(d_changed, 0, nc * sizeof(int), f->stream));
  k_check_pool<<<fblocks((long)total, 256), 256, 0, f->stream>>>(m->g, m->occ_bits.p, m->unk_bits.p, f->pool, d_off,
                                                                d_start, (int)nc, total, d_changed);
  FDBG("k_check_pool");
  k_clear_pool<<<fblocks((long)total, 256), 256, 0, f->stream>>>(f->flag.p, f->pool, d_off, d_start, (int)nc, total,
                                                                d_changed);
  FDBG("k_clear_pool");
  HIPCHK(hipMemcpyAsync(f->h_changed, d_changed, nc * sizeof(int), hipMemcpyDeviceToHost, f->stream));
  // (off / start are pageable: their uploads were staged by the runtime before hipMemcpyAsync returned)
  return FUELMI_OK;
}
// after the stream has drained: apply the verdicts.  removed_ids_ semantics (:74-85): index in
// frontiers_ as the list shrinks; dormant clusters are dropped silently.
static void remove_changed_end(fuelmi_frontier* f) {
  int erased_active = 0;
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
  k_load_var<<<1, 64, 0, f->stream>>>(f->h_var, f->d_var);
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
    static const int ccl_threads = getenv("FUELMI_CCL_THREADS") ? atoi(getenv("FUELMI_CCL_THREADS")) : 512;
    if (ccl_threads == 512)
      k_ccl_local<512><<<f->ccl_tiles, 512, f->ccl_lds, f->stream>>>(g, F, f->TX, f->TY);
    else
      k_ccl_local<256><<<f->ccl_tiles, 256, f->ccl_lds, f->stream>>>(g, F, f->TX, f->TY);
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

// searchFrontiers, first half: drops changed clusters and enqueues the whole device pipeline on the
// frontier's own stream (asynchronous).  The caller may queue other work of the cycle (inflation,
// ESDF, B-spline evaluation on the map's stream) before collecting the result with _search_end.
extern "C" int fuelmi_frontier_search_begin(fuelmi_frontier* f) {
  ARGCHK(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_search_begin");
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  const Geo& g = m->g;
  FArgs& F = f->F;
  f->tmp.clear();
  double umin[3], umax[3];
  fuelmi_map_get_updated_box(m, umin, umax, 1);

  // the scan reads only the occupancy state planes: it runs on its own stream, ordered after the
  // last kernel that rewrote them (fusion / upload), and overlaps the inflation / ESDF / B-spline
  // kernels the caller has queued on the map's stream for the same cycle
  HIPCHK(hipStreamWaitEvent(f->stream, m->ev_planes, 0));
  f->scope.reset(new StageScope(m, FUELMI_K_FRONTIER, f->stream));
  f->removed_ids.clear();
  for (int q = 0; q < 3; ++q) f->rm_lo[q] = 1, f->rm_hi[q] = 0;
  int rc = remove_changed_begin(f, umin, umax);
  if (rc) return rc;

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
  f->search_empty = empty;
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
  if (have_q) {
    const int qx = hv.qreg.hi[0] - hv.qreg.lo[0] + 1, qy = hv.qreg.hi[1] - hv.qreg.lo[1] + 1;
    hv.nty = (qy + f->TY - 1) / f->TY;
    hv.ntiles = ((qx + f->TX - 1) / f->TX) * hv.nty;  // <= f->ccl_tiles (tiles of the whole Q box)
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
  *f->h_var = hv;  // pinned; the previous search has been collected (_search_end synchronises)

  // the second radix pass is needed only beyond 256 kept clusters: guess from the previous search
  f->npass = f->last_nkept > 192 ? 2 : 1;
  f->nb_launch = 256;  // the multisplit kernels stride over however many 2048-cell chunks there are

  // The chain is ~23 dependent launches whose arguments never change (everything per-search sits
  // behind F.var): replay it as a hipGraph -- the host-side launch cost of the individual kernels
  // (~6 us each) was longer than the kernels themselves.
  static const bool no_graph = getenv("FUELMI_NO_GRAPH") != nullptr || getenv("FUELMI_DEBUG_SYNC") != nullptr;
  if (no_graph) return frontier_enqueue_chain(f, f->npass);
  hipGraphExec_t& exec = f->graph_exec[f->npass - 1];
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
  // ~200 us long, an interrupt-driven wake-up costs a noticeable fraction of that
  {
    hipError_t q;
    while ((q = hipStreamQuery(f->stream)) == hipErrorNotReady) {
    }
    HIPCHK(q);
  }
  remove_changed_end(f);
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
  const bool ref_order = f->cfg.reference_order != 0;
  int fin = nkept <= 256 ? 1 : 0;  // buffer pair holding the grouped cells of the search
  std::vector<u32> off2;
  u32 n_in = n_out;
  if (ref_order) {  // cells of every cluster in expandFrontier's order (an NQ seed first), into the other pair
    int rc = frontier_reference_order(f, nq, nkept, n_out, fin, &n_in, &off2);
    if (rc) return rc;
    fin = 1 - fin;
    ncells = n_in;
  }
  if (split_mode) {  // splitLargeFrontiers (:120) on the device; the pinned buffers then hold the pieces
    int rc = frontier_split_run(f, nq, nkept, n_in, fin, &ncl, &ncells, &filtered);
    if (rc) return rc;
    fin = ncl <= 256 ? 1 : 0;  // where the regrouping of the pieces ended
  } else if (ref_order) {
    HIPCHK(hipMemcpyAsync(F.h_cells, F.ms_val[fin], (size_t)n_in * sizeof(u32), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipStreamSynchronize(f->stream));
  }
  f->last_fin = fin;  // buffer holding the grouped cells the lazy clusters point into
  const u32 nchunk = (ref_order && !split_mode) ? 0u : (ncells + SZ_CH - 1) / SZ_CH;  // (records of the grouped
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
  return FUELMI_OK;
}


extern "C" int fuelmi_frontier_search(fuelmi_frontier* f, int* n_new) {
  ARGCHK(f && n_new);
  int rc = fuelmi_frontier_search_begin(f);
  if (rc) return rc;
  return fuelmi_frontier_search_end(f, n_new);
}

extern "C" int fuelmi_frontier_reset(fuelmi_frontier* f) {
  ARGCHK(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_reset");
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  f->frontiers.clear();
  f->dormant.clear();
  f->tmp.clear();
  f->removed_ids.clear();
  f->dirty_all = true;
  f->pool_used = 0;
  k_zero_words<<<fblocks(m->g.W, 256, 1024), 256, 0, f->stream>>>(f->flag.p, m->g.W);
  FDBG("k_zero_words");
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_commit(fuelmi_frontier* f, int dormant) {
  ARGCHK(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_commit");
  auto& dst = dormant ? f->dormant : f->frontiers;
  HIPCHK(hipSetDevice(f->map->device));
  const int rc = frontier_keep_clusters(f, f->tmp);
  if (rc) return rc;
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
  return (int)c->size();
}
extern "C" int fuelmi_frontier_cluster_cells(const fuelmi_frontier* f, int which, int k, int* adr) {
  ARGCHK(f && adr);
  const HCluster* c = nth(f, which, k);
  ARGCHK(c);
  c->copy_to(adr);
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
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  long n = m->g.N;
  int rc = frontier_ensure_stage(f, (size_t)n);
  if (rc) return rc;
  k_expand_flag_bits<<<fblocks(n, 256), 256, 0, f->stream>>>(f->flag.p, n, (char*)f->d_stage);
  FDBG("k_expand_flag_bits");
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(flags, f->d_stage, (size_t)n, hipMemcpyDeviceToHost, f->stream));
  HIPCHK(hipStreamSynchronize(f->stream));
  return FUELMI_OK;
}
