// frontier_order.hip -- the reference's cell ORDER inside a cluster (fuelmi_frontier_cfg.reference_order).
//
// FrontierFinder::expandFrontier (active_perception/src/frontier_finder.cpp:123-164) is a queue BFS: it pops a
// cell, walks its 26 neighbours in the order of allNeighbors (:848-860: x outer, y, z inner) and appends every
// neighbour that is an unflagged frontier cell inside the box with z >= 0.4.  cells_ is the order of discovery.
// Two things depend on that order in their last bits: the sequential f64 sum behind
// average_ (:374-390) and the float accumulation of the pcl::VoxelGrid centroids (:757-774).
//
// The set of cells of a cluster is already known here (frontier.hip); what is computed is each cell's position
// in the queue.  A queue BFS discovers cells level by level (all cells at distance L from the start are popped
// before any at distance L + 1), a cell is discovered by the FIRST popped cell adjacent to it, and the children of
// one popped cell are appended in neighbour order.  Hence
//     key(c) = min over adjacent cells p of the previous level ( rank(p) * 27 + neighbour index(p -> c) )
// orders a level exactly like the queue does.  One workgroup per cluster sweeps the levels:
//   A  every cell p of the current level proposes its key to each member neighbour (atomicMin; keys of later
//      levels are larger than every key handed out before, so cells that are already placed ignore them)
//   B  every p counts the neighbours whose final key is its own proposal (its children), a prefix sum over the
//      level gives their positions, and they are written in neighbour order: the next level, already sorted.
// The number of levels is the graph eccentricity of the start cell (hundreds for a surface that spans the map), and
// a level is a handful of DEPENDENT steps, so what a level costs is the latency of the memory the sweep lives in:
//   k_bfs_nbr + k_bfs_sweep  clusters of at most BFSL_CAP cells (every cluster of an incremental search).  The part
//                    that does not depend on the order -- which of a cell's 26 neighbours are cells of the cluster,
//                    and where in the cluster's sorted address list they are -- is found for all cells at once by
//                    k_bfs_nbr (a lane per cell, nine lower-bound searches side by side) and left as a 24-byte
//                    record per cell.  k_bfs_sweep then runs the levels with the keys and the queue in LDS (and
//                    the records too when they fit): a level is one record fetch, LDS atomics and three barriers.
//   k_bfs_sweep_g + k_bfs_emit  larger clusters (round 4; the round-2 kernel walked the chain's per-voxel records and
//                    paid ~25 us per level): the same neighbour records with 32-bit list indices (48 bytes per cell), the
//                    keys in a dense array by list index (4 bytes per cell: 0.56 MB for the 139 k cells of a surface
//                    that spans the 400 x 400 x 100 map -- L2-resident), the queue in global memory with its tail
//                    mirrored in an LDS ring.  A level is one record fetch, atomics that do not return, one round of
//                    key reads and three barriers: ~2-3 us.  The ordered addresses are written by a grid-wide kernel.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "frontier_internal.h"

struct OrderScratch {
  u32* key = nullptr;     // [big_cap] discovery keys by grouped position (large clusters; set to "unset" by k_bfs_nbr)
  u32* ord = nullptr;     // [cap_q + cap_kept] list indices in BFS order (large clusters)
  u32* nbr32 = nullptr;   // [big_cap][12] neighbour records with 32-bit list indices (large clusters)
  size_t big_cap = 0;
  u32* h_err = nullptr;   // pinned [4], written by the kernels
  uint2* nbr = nullptr;   // [nbr_cap][3] neighbour records of k_bfs_nbr (grouped cells, then one per cluster for its seed)
  size_t nbr_cap = 0;
  u32* first = nullptr;   // [cap_kept] index of the start cell in its cluster's list
  bool lds_attr = false;  // k_bfs_sweep may use the whole LDS of a CU
};

namespace {

#define BFS_T 1024
#define BFSL_T 256
#define BFSL_CAP FR_REFORDER_AUTO  // cells of a cluster ordered inside LDS (6 bytes each: key, queue entry)
#define BFSL_LDS_MAX (160u * 1024u - 64u)
#define BFSG_RING 32768u  // queue entries mirrored in LDS (128 KB)

struct BArgs {
  u32* key;
  u32* ord;
  const u32* in_adr;  // grouped cells of the search (ascending address inside a cluster), cluster r at krec[r].off
  u32* out_adr;
  u32* out_key;
  u32* h_out;         // pinned host copy of out_adr (nullptr: the caller does not want one)
  u32 nq;
  u32 lcap;           // largest cluster of this search that k_bfs_sweep orders
  uint2* nbr;         // neighbour records
  u32* nbr32;         // ... of the large clusters
  u32* first;
  u32 n_grouped;      // cells in the grouped array; record n_grouped + r belongs to the seed of cluster r
  u32 nkept;
  int nbr_in_lds;     // the records of a cluster fit into LDS beside its keys and its queue
  u32* err;
};

// first position of cluster r in the ordered array: the sizes of the clusters before it (a size counts the NQ seed
// that started the cluster, the grouped array does not hold it)
template <int T>
__device__ __forceinline__ u32 cluster_base(const FArgs& F, u32 r, u32* s_red) {
  u32 v = 0u;
  for (u32 k = threadIdx.x; k < r; k += T) v += F.krec[k].size;
  for (int off = 32; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  u32 tot = 0u;
  for (int w = 0; w < T / 64; ++w) tot += s_red[w];
  __syncthreads();
  return tot;
}

// workgroup scope: what one workgroup's waves exchange through global memory between its own barriers (k_bfs_sweep_g).
// Agent scope would send every atomic and every load past the XCD's L2 to the memory side (multi-XCD coherence):
// ~8 us per BFS level instead of ~3.
__device__ __forceinline__ u32 ld_wg(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st_wg(u32* p, u32 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- clusters that fit into LDS ----
// What k_bfs_nbr leaves per cell: which cells of the 3 x 3 x 3 block around it are in the cluster's list (3 bits per
// (dx, dy) line: z-1, z, z+1) and the list index of every line's first present cell.
struct NbrSet {
  u32 raw;                        // bits 0..26 as above; bit 27: the cell has z == 0, bit 28: z == nz - 1
  u32 lo8;                        // index of the first present cell of line 8
  unsigned long long lo03, lo47;  // ... of lines 0..3 and 4..7, 16 bits each (packed: a lane picks a line by a
                                  // computed number, and an indexed array would live in scratch)
};

// the neighbours a cell proposes to: not itself, and not across the z ends of a line (bit 0 of a line at z == 0 is
// the last cell of the previous line, bit 2 at z == nz - 1 the first of the next)
__device__ __forceinline__ u32 nbr_valid(const NbrSet& s) {
  u32 ok = s.raw & 0x7FFFFFFu & ~(2u << 12);
  if (s.raw & (1u << 27)) ok &= ~0x1249249u;
  if (s.raw & (1u << 28)) ok &= ~(0x1249249u << 2);
  return ok;
}

// list index of neighbour idx27 = 3 * line + b
__device__ __forceinline__ u32 nbr_index(const NbrSet& s, int idx27) {
  const int li = idx27 / 3, b = idx27 - 3 * li;
  const u32 below = (s.raw >> (3 * li)) & ((1u << b) - 1u);
  const unsigned long long w = li < 4 ? s.lo03 : s.lo47;
  const u32 first = li == 8 ? s.lo8 : (u32)(w >> (16 * (li & 3))) & 0xFFFFu;
  return first + (u32)__popc(below);
}

// the block of cell a in the sorted list adr[0..n): returns the 27 presence bits (+ the z-end flags), lo[li] = list
// index of the first cell at or after the start of line li
__device__ __forceinline__ u32 find_neighbours_raw(const Geo& g, const u32* adr, u32 n, int a, u32 (&lo)[9]) {
  const int x = a / g.nyz;
  const int rr = a - x * g.nyz;
  const int y = rr / g.nz, z = rr - y * g.nz;
  u32 top = 1u;
  while (top * 2u <= n) top *= 2u;
  if (n == 0u) top = 0u;
  int t[9];
#pragma unroll
  for (int li = 0; li < 9; ++li) {
    const int dx = li / 3 - 1, dy = li % 3 - 1;
    t[li] = a + dx * g.nyz + dy * g.nz - 1;
    lo[li] = 0u;
  }
  for (u32 step = top; step > 0u; step >>= 1) {  // nine lower bounds side by side: lo = cells below t
#pragma unroll
    for (int li = 0; li < 9; ++li) {
      const u32 q = lo[li] + step;
      if (q <= n && (int)adr[q - 1u] < t[li]) lo[li] = q;
    }
  }
  u32 raw = 0u;
#pragma unroll
  for (int li = 0; li < 9; ++li) {
    const int dx = li / 3 - 1, dy = li % 3 - 1;
    const bool inside = (unsigned)(x + dx) < (unsigned)g.nx && (unsigned)(y + dy) < (unsigned)g.ny;
    u32 q = lo[li], bits = 0u;
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (q < n && (int)adr[q] == t[li] + b) bits |= 1u << b, ++q;
    raw |= (inside ? bits : 0u) << (3 * li);
  }
  if (z == 0) raw |= 1u << 27;
  if (z == g.nz - 1) raw |= 1u << 28;
  return raw;
}
__device__ __forceinline__ NbrSet find_neighbours(const Geo& g, const u32* adr, u32 n, int a) {
  NbrSet s;
  u32 lo[9];
  s.raw = find_neighbours_raw(g, adr, n, a, lo);
  s.lo03 = (unsigned long long)lo[0] | (unsigned long long)lo[1] << 16 | (unsigned long long)lo[2] << 32 |
           (unsigned long long)lo[3] << 48;
  s.lo47 = (unsigned long long)lo[4] | (unsigned long long)lo[5] << 16 | (unsigned long long)lo[6] << 32 |
           (unsigned long long)lo[7] << 48;
  s.lo8 = lo[8];
  return s;
}

// large clusters: the same record with 32-bit indices, 12 words (48 bytes, three 16-byte accesses; word 10, 11 unused)
struct Nbr32 {
  u32 raw;
  u32 lo[9];  // (only ever indexed by compile-time constants: registers)
};
__device__ __forceinline__ u32 nbr32_valid(u32 raw) {
  u32 ok = raw & 0x7FFFFFFu & ~(2u << 12);
  if (raw & (1u << 27)) ok &= ~0x1249249u;
  if (raw & (1u << 28)) ok &= ~(0x1249249u << 2);
  return ok;
}
__device__ __forceinline__ Nbr32 nbr32_load(const u32* rec) {
  const uint4 a = reinterpret_cast<const uint4*>(rec)[0], b = reinterpret_cast<const uint4*>(rec)[1],
              c = reinterpret_cast<const uint4*>(rec)[2];
  Nbr32 s;
  s.raw = a.x, s.lo[0] = a.y, s.lo[1] = a.z, s.lo[2] = a.w;
  s.lo[3] = b.x, s.lo[4] = b.y, s.lo[5] = b.z, s.lo[6] = b.w;
  s.lo[7] = c.x, s.lo[8] = c.y;
  return s;
}
__device__ __forceinline__ void nbr_store(uint2* rec, const NbrSet& s) {
  rec[0] = make_uint2(s.raw, s.lo8);
  rec[1] = make_uint2((u32)s.lo03, (u32)(s.lo03 >> 32));
  rec[2] = make_uint2((u32)s.lo47, (u32)(s.lo47 >> 32));
}
__device__ __forceinline__ NbrSet nbr_load(const uint2* rec) {
  const uint2 a = rec[0], b = rec[1], c = rec[2];
  NbrSet s;
  s.raw = a.x, s.lo8 = a.y;
  s.lo03 = (unsigned long long)b.x | (unsigned long long)b.y << 32;
  s.lo47 = (unsigned long long)c.x | (unsigned long long)c.y << 32;
  return s;
}

// a lane per grouped cell (then a lane per cluster for its NQ seed): the cell's neighbour record
__global__ void __launch_bounds__(256) k_bfs_nbr(Geo g, FArgs F, BArgs B) {
  const u32 p = blockIdx.x * 256u + threadIdx.x;
  if (p >= B.n_grouped + B.nkept) return;
  u32 r;
  bool is_seed = false;
  if (p < B.n_grouped) {
    u32 lo = 0u, top = 1u;  // the cluster whose range holds position p: the last with off <= p
    while (top * 2u <= B.nkept) top *= 2u;
    for (u32 step = top; step > 0u; step >>= 1)
      if (lo + step < B.nkept && F.krec[lo + step].off <= p) lo += step;
    r = lo;
  } else {
    r = p - B.n_grouped;
    is_seed = true;
  }
  const KeptRec kr = F.krec[r];
  const bool seedc = kr.slot >= B.nq;
  const u32 n = kr.size - (seedc ? 1u : 0u);
  if (is_seed ? !seedc : p - kr.off >= n) return;
  const u32* list = B.in_adr + kr.off;
  const u32 a = is_seed ? kr.addr : list[p - kr.off];
  if (kr.size > BFSL_CAP) {  // (k_bfs_sweep_g's)
    u32 lo[9];
    const u32 raw = find_neighbours_raw(g, list, n, (int)a, lo);
    uint4* rec = reinterpret_cast<uint4*>(B.nbr32 + 12 * (size_t)p);
    rec[0] = make_uint4(raw, lo[0], lo[1], lo[2]);
    rec[1] = make_uint4(lo[3], lo[4], lo[5], lo[6]);
    rec[2] = make_uint4(lo[7], lo[8], 0u, 0u);
    B.key[p] = 0xFFFFFFFFu;
  } else {
    nbr_store(B.nbr + 3 * (size_t)p, find_neighbours(g, list, n, (int)a));
  }
  if (is_seed)
    B.first[r] = n;
  else if (!seedc && a == kr.addr)
    B.first[r] = p - kr.off;
}

// one workgroup per cluster: the level sweep.  key[i] = discovery key of cell i of the cluster's list (i == n: the NQ
// seed, reachable by nobody), ord[k] = list index of the k-th cell of the queue.
__global__ void __launch_bounds__(BFSL_T) k_bfs_sweep(Geo g, FArgs F, BArgs B) {
  extern __shared__ __align__(16) u32 bfs_lds[];
  __shared__ u32 s_wave[BFSL_T / 64];
  const u32 r = blockIdx.x;
  const KeptRec kr = F.krec[r];
  const u32 want = kr.size;
  if (want > BFSL_CAP) return;  // (k_bfs_order's)
  const bool seedc = kr.slot >= B.nq;
  const u32 n = want - (seedc ? 1u : 0u);
  u32* key = bfs_lds;                                                                 // [lcap + 1]
  uint2* tab = reinterpret_cast<uint2*>(bfs_lds + ((B.lcap + 2u) & ~1u));             // [lcap + 1][3] (optional)
  unsigned short* ord = reinterpret_cast<unsigned short*>(
      bfs_lds + ((B.lcap + 2u) & ~1u) + (B.nbr_in_lds ? 6u * (B.lcap + 1u) : 0u));  // [lcap + 1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint2* grec = B.nbr + 3 * (size_t)kr.off;
  const uint2* srec = B.nbr + 3 * ((size_t)B.n_grouped + r);
  for (u32 k = threadIdx.x; k <= n; k += BFSL_T) key[k] = 0xFFFFFFFFu;
  if (B.nbr_in_lds) {
    for (u32 k = threadIdx.x; k < 3u * n; k += BFSL_T) tab[k] = grec[k];
    if (seedc && threadIdx.x < 3) tab[3u * n + threadIdx.x] = srec[threadIdx.x];
  }
  const u32 base0 = cluster_base<BFSL_T>(F, r, s_wave);  // (barriers inside)
  if (threadIdx.x == 0) {
    u32 first = B.first[r];
    if (first > n || (first == n) != seedc) B.err[2] = 1u + r, first = 0u;  // the claimer is a cell of its cluster
    key[first] = 0u;
    ord[0] = (unsigned short)first;
  }
  __syncthreads();
  auto record = [&](u32 ci) {
    if (B.nbr_in_lds) return nbr_load(tab + 3u * ci);
    return nbr_load(ci < n ? grec + 3u * ci : srec);
  };
  u32 lev_lo = 0u, lev_hi = 1u;  // positions of the current level in the queue (uniform)
  u32 n_lev = 0u;
  bool bad = false;
  while (true) {
    const u32 nL = lev_hi - lev_lo;
    ++n_lev;
    NbrSet keep;  // the record of this lane's first cell of the level, fetched once for both passes
    keep.raw = keep.lo8 = 0u, keep.lo03 = keep.lo47 = 0ull;
    // ---- A: proposals ----
    for (u32 j = threadIdx.x; j < nL; j += BFSL_T) {
      const NbrSet s = record(ord[lev_lo + j]);
      if (j < BFSL_T) keep = s;
      const u32 kbase = (lev_lo + j) * 27u + 1u;
      u32 m = nbr_valid(s);
      while (m) {
        const int idx27 = __builtin_ctz(m);
        m &= m - 1u;
        atomicMin(&key[nbr_index(s, idx27)], kbase + (u32)idx27);
      }
    }
    __syncthreads();
    // ---- B: children of every cell of the level, placed in (parent, neighbour) order ----
    u32 run = 0u;
    for (u32 c0 = 0u; c0 < nL; c0 += BFSL_T) {
      const u32 j = c0 + threadIdx.x;
      u32 wmask = 0u;
      NbrSet s = keep;
      if (j < nL) {
        if (c0) s = record(ord[lev_lo + j]);
        const u32 kbase = (lev_lo + j) * 27u + 1u;
        const u32 ok = nbr_valid(s);
        // all 27 keys are fetched before the first is looked at (a loop over the set bits would wait for every
        // LDS read in turn: ~1 us of the ~3 us a level takes); absent neighbours read key[0] and are masked
        u32 kv[27];
#pragma unroll
        for (int li = 0; li < 9; ++li) {
          const u32 first = li == 8 ? s.lo8 : (u32)((li < 4 ? s.lo03 : s.lo47) >> (16 * (li & 3))) & 0xFFFFu;
          const u32 bits = (s.raw >> (3 * li)) & 7u;
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const u32 idx = first + (u32)__popc(bits & ((1u << b) - 1u));
            kv[3 * li + b] = key[(ok >> (3 * li + b)) & 1u ? idx : 0u];
          }
        }
#pragma unroll
        for (int i27 = 0; i27 < 27; ++i27)
          if (kv[i27] == kbase + (u32)i27) wmask |= ok & (1u << i27);
      }
      const u32 cnt = (u32)__popc(wmask);
      // inclusive scan of the counts (<= 26 each) over the wave, one bit plane at a time: ballots and mbcnt, no
      // cross-lane data movement
      u32 v = cnt;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const unsigned long long bal = __ballot((cnt >> b) & 1u);
        v += (u32)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)) << b;
      }
      if (c0) __syncthreads();  // (the sums of the previous chunk have been read)
      if (lane == 63) s_wave[wave] = v;
      __syncthreads();
      u32 woff = 0u, total = 0u;
      for (int w = 0; w < BFSL_T / 64; ++w) {
        if (w < wave) woff += s_wave[w];
        total += s_wave[w];
      }
      u32 pos = lev_hi + run + woff + (v - cnt);
      while (wmask) {
        const int idx27 = __builtin_ctz(wmask);
        wmask &= wmask - 1u;
        if (pos < want)
          ord[pos] = (unsigned short)nbr_index(s, idx27);
        else
          bad = true;  // more cells reached than the cluster holds: cannot happen
        ++pos;
      }
      run += total;
    }
    __syncthreads();
    if (run == 0u) break;
    lev_lo = lev_hi;
    lev_hi += run;
    if (lev_hi > want) break;
  }
  if (bad) B.err[0] = 1u;
  if (lev_hi != want && threadIdx.x == 0) B.err[1] = 1u + r;  // the sweep did not reach every cell of the cluster
  if (threadIdx.x == 0 && r == 0u) B.err[3] = n_lev;  // (diagnostics: FUELMI_FR_TIMING)
  for (u32 k = threadIdx.x; k < want && k < lev_hi; k += BFSL_T) {
    const u32 ci = ord[k];
    const u32 a = ci < n ? B.in_adr[kr.off + ci] : kr.addr;
    B.out_adr[base0 + k] = a;
    B.out_key[base0 + k] = r;
    if (B.h_out) B.h_out[base0 + k] = a;
  }
}

// ---- larger clusters: keys and queue in global memory (L2), the records of k_bfs_nbr with 32-bit indices ----
// one workgroup per cluster.  key[i]: discovery key of cell i of the cluster's list (the NQ seed has none: nobody
// reaches it), ordg[k] = list index of the k-th cell of the queue (n = the seed), mirrored in the LDS ring.
template <int BFSG_T>
__global__ void __launch_bounds__(BFSG_T) k_bfs_sweep_g(Geo g, FArgs F, BArgs B) {
  extern __shared__ __align__(16) u32 ring[];
  __shared__ u32 s_wave[BFSG_T / 64];
  const u32 r = blockIdx.x;
  const KeptRec kr = F.krec[r];
  const u32 want = kr.size;
  if (want <= BFSL_CAP) return;  // (k_bfs_sweep's)
  const bool seedc = kr.slot >= B.nq;
  const u32 n = want - (seedc ? 1u : 0u);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32* key = B.key + kr.off;
  const u32* grec = B.nbr32 + 12 * (size_t)kr.off;
  const u32* srec = B.nbr32 + 12 * ((size_t)B.n_grouped + r);
  const u32 base0 = cluster_base<BFSG_T>(F, r, s_wave);  // (barriers inside)
  u32* ordg = B.ord + base0;
  if (threadIdx.x == 0) {
    u32 first = B.first[r];
    if (first > n || (first == n) != seedc) B.err[2] = 1u + r, first = 0u;  // the claimer is a cell of its cluster
    if (first < n) st_wg(&key[first], 0u);
    ring[0] = first;
  }
  __syncthreads();
  // The queue lives in the ring; positions [0, flushed) have been copied to ordg.  A level whose children might wrap onto
  // positions not copied yet ("direct") first copies what is pending and writes its children to ordg as well.
  u32 lev_lo = 0u, lev_hi = 1u;  // positions of the current level in the queue (uniform)
  u32 flushed = 0u;
  u32 n_lev = 0u;
  bool bad = false;
  auto flush_to = [&](u32 upto) {  // (every position in [flushed, upto) is intact in the ring: see `direct`)
    for (u32 k = flushed + threadIdx.x; k < upto; k += BFSG_T) st_wg(&ordg[k], ring[k & (BFSG_RING - 1u)]);
    flushed = upto;
  };
  while (true) {
    const u32 nL = lev_hi - lev_lo;
    ++n_lev;
    const u32 maxrun = min(26u * nL, want - lev_hi);
    const bool direct = lev_hi + maxrun - flushed > BFSG_RING;
    // the level is read from the ring when nothing written during this level can wrap onto it
    const bool in_ring = nL + maxrun <= BFSG_RING;
    if (direct) {
      flush_to(lev_hi);
      if (!in_ring) __syncthreads();  // (the level is about to be read back from ordg)
    }
    auto cell_at = [&](u32 pos) { return in_ring ? ring[pos & (BFSG_RING - 1u)] : ld_wg(&ordg[pos]); };
    Nbr32 keep;  // the record of this lane's first cell of the level, fetched once for both passes
    keep.raw = 0u;
#pragma unroll
    for (int li = 0; li < 9; ++li) keep.lo[li] = 0u;
    // ---- A: proposals ----
    for (u32 j = threadIdx.x; j < nL; j += BFSG_T) {
      const u32 ci = cell_at(lev_lo + j);
      const Nbr32 s = nbr32_load(ci < n ? grec + 12 * (size_t)ci : srec);
      if (j < BFSG_T) keep = s;
      const u32 kbase = (lev_lo + j) * 27u + 1u;
      const u32 ok = nbr32_valid(s.raw);
#pragma unroll
      for (int li = 0; li < 9; ++li) {
        const u32 bits = (s.raw >> (3 * li)) & 7u;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if ((ok >> (3 * li + b)) & 1u)
            (void)__hip_atomic_fetch_min(&key[s.lo[li] + (u32)__popc(bits & ((1u << b) - 1u))], kbase + (u32)(3 * li + b),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    // ---- B: children of every cell of the level, placed in (parent, neighbour) order ----
    u32 run = 0u;
    for (u32 c0 = 0u; c0 < nL; c0 += BFSG_T) {
      const u32 j = c0 + threadIdx.x;
      u32 wmask = 0u;
      Nbr32 s = keep;
      if (j < nL) {
        if (c0) {
          const u32 ci = cell_at(lev_lo + j);
          s = nbr32_load(ci < n ? grec + 12 * (size_t)ci : srec);
        }
        const u32 kbase = (lev_lo + j) * 27u + 1u;
        const u32 ok = nbr32_valid(s.raw);
        u32 kv[27];  // the keys of the present neighbours, all requested before the first is looked at
#pragma unroll
        for (int li = 0; li < 9; ++li) {
          const u32 bits = (s.raw >> (3 * li)) & 7u;
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const u32 idx = s.lo[li] + (u32)__popc(bits & ((1u << b) - 1u));
            kv[3 * li + b] = 0u;
            if ((ok >> (3 * li + b)) & 1u) kv[3 * li + b] = ld_wg(&key[idx]);
          }
        }
#pragma unroll
        for (int i27 = 0; i27 < 27; ++i27)
          if (kv[i27] == kbase + (u32)i27) wmask |= ok & (1u << i27);
      }
      const u32 cnt = (u32)__popc(wmask);
      u32 v = cnt;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const unsigned long long bal = __ballot((cnt >> b) & 1u);
        v += (u32)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)) << b;
      }
      if (c0) __syncthreads();  // (the sums of the previous chunk have been read)
      if (lane == 63) s_wave[wave] = v;
      __syncthreads();
      u32 woff = 0u, total = 0u;
      for (int w = 0; w < BFSG_T / 64; ++w) {
        if (w < wave) woff += s_wave[w];
        total += s_wave[w];
      }
      u32 pos = lev_hi + run + woff + (v - cnt);
      if (wmask) {
#pragma unroll
        for (int li = 0; li < 9; ++li) {
          const u32 bits = (s.raw >> (3 * li)) & 7u;
#pragma unroll
          for (int b = 0; b < 3; ++b)
            if ((wmask >> (3 * li + b)) & 1u) {
              const u32 idx = s.lo[li] + (u32)__popc(bits & ((1u << b) - 1u));
              if (pos < want) {
                ring[pos & (BFSG_RING - 1u)] = idx;
                if (direct) st_wg(&ordg[pos], idx);
              } else {
                bad = true;  // more cells reached than the cluster holds: cannot happen
              }
              ++pos;
            }
        }
      }
      run += total;
    }
    __syncthreads();
    if (direct) flushed = min(lev_hi + run, want);
    if (run == 0u) break;
    lev_lo = lev_hi;
    lev_hi += run;
    if (lev_hi > want) break;
  }
  if (lev_hi <= want) flush_to(lev_hi);
  if (bad) B.err[0] = 1u;
  if (lev_hi != want && threadIdx.x == 0) B.err[1] = 1u + r;  // the sweep did not reach every cell of the cluster
  if (threadIdx.x == 0) B.err[3] = n_lev;  // (diagnostics: FUELMI_FR_TIMING)
}

// ordered addresses + cluster rank of every cell of the large clusters (grid: chunks x clusters)
__global__ void __launch_bounds__(256) k_bfs_emit(Geo g, FArgs F, BArgs B) {
  __shared__ u32 s_red[4];
  const u32 r = blockIdx.y;
  const KeptRec kr = F.krec[r];
  const u32 want = kr.size;
  if (want <= BFSL_CAP) return;
  const u32 n = want - (kr.slot >= B.nq ? 1u : 0u);
  const u32 base0 = cluster_base<256>(F, r, s_red);
  for (u32 k = blockIdx.x * 256u + threadIdx.x; k < want; k += gridDim.x * 256u) {
    const u32 ci = B.ord[base0 + k];
    const u32 a = ci < n ? B.in_adr[kr.off + ci] : kr.addr;
    B.out_adr[base0 + k] = a;
    B.out_key[base0 + k] = r;
    if (B.h_out) B.h_out[base0 + k] = a;
  }
}

}  // namespace

void frontier_order_free(fuelmi_frontier* f) {
  OrderScratch* o = f->order;
  if (!o) return;
  void* dev[] = {o->key, o->ord, o->nbr, o->nbr32, o->first};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (o->h_err) (void)hipHostFree(o->h_err);
  delete o;
  f->order = nullptr;
}

int frontier_reference_order(fuelmi_frontier* f, u32 nq, u32 nkept, u32 n_out, int fin, u32* n_total,
                             std::vector<u32>* h_off2, bool fetch_cells) {
  fuelmi_map* m = f->map;
  FArgs& F = f->F;
  hipStream_t st = f->stream;
  if (!f->order) {
    OrderScratch* o = new OrderScratch;
    f->order = o;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&o->h_err), 16 * sizeof(u32), hipHostMallocDefault));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->first), (size_t)F.cap_kept * sizeof(u32)));
  }
  OrderScratch* o = f->order;
  // cluster r occupies off2[r] .. off2[r+1]: its Q0 cells plus the NQ seed that started it, if one did
  std::vector<u32>& off2 = *h_off2;
  off2.assign(nkept + 1, 0u);
  u32 lcap = 0u, n_grouped = 0u;
  bool any_big = false;
  for (u32 r = 0; r < nkept; ++r) {
    const u32 sz = F.h_rec[r].size;  // size counts the seed
    off2[r + 1] = off2[r] + sz;
    n_grouped = std::max(n_grouped, F.h_rec[r].off + sz - (F.h_rec[r].slot >= nq ? 1u : 0u));
    if (sz <= BFSL_CAP)
      lcap = std::max(lcap, sz);
    else
      any_big = true;
  }
  const u32 total = off2[nkept];
  if (total > F.cap_q) {
    fuelmi_set_error("reference order: %u cells exceed the capacity %u", total, F.cap_q);
    return FUELMI_ELIMIT;
  }
  (void)n_out;
  const bool timing = getenv("FUELMI_FR_TIMING") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  BArgs B;
  B.key = o->key, B.ord = o->ord;
  B.in_adr = F.ms_val[fin];
  B.out_adr = F.ms_val[1 - fin], B.out_key = F.ms_key[1 - fin];
  B.h_out = fetch_cells ? F.h_cells : nullptr;  // (written by the kernels themselves: no copy engine, no second wait)
  B.nq = nq, B.lcap = lcap, B.err = o->h_err;
  B.nbr = nullptr, B.nbr32 = nullptr, B.first = o->first, B.n_grouped = n_grouped, B.nkept = nkept, B.nbr_in_lds = 0;
  for (int k = 0; k < 4; ++k) o->h_err[k] = 0u;  // (the kernels of the previous search are long done)
  const size_t need = (size_t)n_grouped + nkept;
  if (lcap && need > o->nbr_cap) {  // (grown by need: 24 bytes per cell of the largest search so far)
    if (o->nbr) HIPCHK(hipFree(o->nbr));
    o->nbr = nullptr, o->nbr_cap = 0;
    const size_t cap = std::max<size_t>(need + need / 2, 1u << 16);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->nbr), cap * 3 * sizeof(uint2)));
    o->nbr_cap = cap;
  }
  if (any_big && need > o->big_cap) {  // (52 bytes per cell: records and keys)
    if (o->nbr32) HIPCHK(hipFree(o->nbr32));
    if (o->key) HIPCHK(hipFree(o->key));
    o->nbr32 = o->key = nullptr, o->big_cap = 0;
    const size_t cap = std::max<size_t>(need + need / 2, 1u << 16);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->nbr32), cap * 12 * sizeof(u32)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->key), cap * sizeof(u32)));
    o->big_cap = cap;
  }
  if (any_big && !o->ord) HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->ord), ((size_t)F.cap_q + F.cap_kept) * sizeof(u32)));
  B.nbr = o->nbr, B.nbr32 = o->nbr32, B.key = o->key, B.ord = o->ord;
  k_bfs_nbr<<<(u32)((need + 255) / 256), 256, 0, st>>>(m->g, F, B);
  HIPCHK(hipGetLastError());
  if (lcap) {
    const size_t base = (((size_t)lcap + 2) & ~(size_t)1) * 4 + ((size_t)lcap + 1) * 2 + 16;
    B.nbr_in_lds = base + ((size_t)lcap + 1) * 24 <= BFSL_LDS_MAX ? 1 : 0;
    const size_t lds = base + (B.nbr_in_lds ? ((size_t)lcap + 1) * 24 : 0);
    if (!o->lds_attr) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bfs_sweep), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)BFSL_LDS_MAX));
      for (const void* fn : {reinterpret_cast<const void*>(&k_bfs_sweep_g<256>), reinterpret_cast<const void*>(&k_bfs_sweep_g<512>),
                             reinterpret_cast<const void*>(&k_bfs_sweep_g<1024>)})
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BFSG_RING * sizeof(u32))));
      o->lds_attr = true;
    }
    k_bfs_sweep<<<nkept, BFSL_T, lds, st>>>(m->g, F, B);
    HIPCHK(hipGetLastError());
  }
  if (any_big) {
    if (!o->lds_attr) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bfs_sweep), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)BFSL_LDS_MAX));
      for (const void* fn : {reinterpret_cast<const void*>(&k_bfs_sweep_g<256>), reinterpret_cast<const void*>(&k_bfs_sweep_g<512>),
                             reinterpret_cast<const void*>(&k_bfs_sweep_g<1024>)})
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BFSG_RING * sizeof(u32))));
      o->lds_attr = true;
    }
    // (512 lanes: measured against 256 and 1 024, profiles/r04_reference_order_timing.txt)
    k_bfs_sweep_g<512><<<nkept, 512, BFSG_RING * sizeof(u32), st>>>(m->g, F, B);
    HIPCHK(hipGetLastError());
    k_bfs_emit<<<dim3(64, nkept), 256, 0, st>>>(m->g, F, B);
    HIPCHK(hipGetLastError());
  }
  for (;;) {  // (poll: a blocking synchronisation adds ~15 us of wake-up to a sweep of a few hundred)
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) HIPCHK(q);
  }
  if (timing)
    std::fprintf(stderr, "[fr-timing] reference order: %u clusters, %u cells, largest in LDS %u (levels of cluster 0: %u), global sweep %d: %.1f us\n",
                 nkept, total, lcap, o->h_err[3], any_big ? 1 : 0,
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
  if (o->h_err[0] || o->h_err[1] || o->h_err[2]) {
    fuelmi_set_error("reference order: the level sweep of cluster %u did not match its cell set",
                     (o->h_err[1] ? o->h_err[1] : o->h_err[2]) - 1u);
    return FUELMI_EHIP;
  }
  *n_total = total;
  return FUELMI_OK;
}
