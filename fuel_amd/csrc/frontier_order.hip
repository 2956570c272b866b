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
// Cost: two passes over ~26 neighbour look-ups per cell and two workgroup barriers per level; the number of
// levels is the graph eccentricity of the start cell (hundreds for a surface that spans the map).
#include <cstdlib>
#include <vector>

#include "frontier_internal.h"

struct OrderScratch {
  u32* key = nullptr;     // [N] discovery keys by voxel address (0xFFFFFFFF outside a running sweep)
  u32* ord = nullptr;     // [cap_q + cap_kept] addresses in BFS order
  u32* off2 = nullptr;    // [cap_kept + 1]
  u32* err = nullptr;     // [4]
  u32* h_err = nullptr;   // pinned [4]
};

namespace {

#define BFS_T 1024

struct BArgs {
  u32* key;
  u32* ord;
  const u32* off2;
  u32* out_adr;
  u32* out_key;
  u32 nq;
  u32* err;
};

__device__ __forceinline__ u32 ld_agent(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u32* p, u32 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// is the Q0 cell at address a = (xx, yy, .) a member of the cluster?  (fast chain: component of the cell by
// address -> code of the component; legacy chain: per-cell claimer slot behind the compact index)
__device__ __forceinline__ bool is_member(const FArgs& F, long a, int xx, int yy, int slot, u32 rank) {
  if (F.fast) {
    const FVar& V = *F.var;
    const int tile = ((xx - V.px0) / V.ftx) * V.nty_f + (yy - V.py0) / V.fty;
    return F.rcode[F.t_base[tile] + (u32)F.vlab[a]] == rank;
  }
  const u32 cj = rank_q(F, a);
  return cj < F.cap_q && F.cell_slot[cj] == slot;
}

// member neighbours of the cell at address a, in allNeighbors order: calls fn(idx27, address)
template <typename Fn>
__device__ __forceinline__ void for_member_neighbours(const Geo& g, const FArgs& F, long a, int slot, u32 rank, Fn fn) {
  const int x = (int)(a / g.nyz);
  const int r = (int)(a - (long)x * g.nyz);
  const int y = r / g.nz, z = r - y * g.nz;
  for (int dx = -1; dx <= 1; ++dx) {
    const int xx = x + dx;
    if (xx < 0 || xx >= g.nx) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= g.ny) continue;
      const long nb0 = a + (long)dx * g.nyz + (long)dy * g.nz - 1;
      // (fast chain: the Q0 bits live in the per-tile segment arrays; legacy chain: in the Q0 plane)
      u32 bits = F.fast ? q_bits3(g, *F.var, F, xx, yy, z) : (u32)(plane_window(F.qb, nb0) & 7ull);
      if (z == 0) bits &= ~1u;
      if (z == g.nz - 1) bits &= ~4u;
      if (dx == 0 && dy == 0) bits &= ~2u;
      while (bits) {
        const int b = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (is_member(F, nb0 + b, xx, yy, slot, rank)) fn((dx + 1) * 9 + (dy + 1) * 3 + b, (u32)(nb0 + b));
      }
    }
  }
}

__global__ void __launch_bounds__(BFS_T) k_bfs_order(Geo g, FArgs F, BArgs B) {
  __shared__ u32 s_wave[BFS_T / 64];
  __shared__ u32 s_run;
  const u32 r = blockIdx.x;
  const KeptRec kr = F.krec[r];
  const int slot = (int)kr.slot;
  const bool seedc = kr.slot >= B.nq;  // started by an NQ seed: the seed is cells_[0] but not a Q0 cell
  const u32 base0 = B.off2[r];
  const u32 want = B.off2[r + 1] - base0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    if (!seedc) st_agent(&B.key[kr.addr], 0u);
    st_agent(&B.ord[base0], kr.addr);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  u32 lev_lo = 0u, lev_hi = 1u;  // positions of the current level inside the cluster (uniform)
  while (true) {
    const u32 nL = lev_hi - lev_lo;
    // ---- A: proposals ----
    for (u32 j = threadIdx.x; j < nL; j += BFS_T) {
      const long a = (long)ld_agent(&B.ord[base0 + lev_lo + j]);
      const u32 kbase = (lev_lo + j) * 27u + 1u;
      for_member_neighbours(g, F, a, slot, r, [&](int idx27, u32 aj) {
        (void)__hip_atomic_fetch_min(&B.key[aj], kbase + (u32)idx27, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- B: children of every cell of the level, placed in (parent, neighbour) order ----
    if (threadIdx.x == 0) s_run = 0u;
    __syncthreads();
    for (u32 c0 = 0u; c0 < nL; c0 += BFS_T) {
      const u32 j = c0 + threadIdx.x;
      u32 wmask = 0u;
      long a = 0;
      if (j < nL) {
        a = (long)ld_agent(&B.ord[base0 + lev_lo + j]);
        const u32 kbase = (lev_lo + j) * 27u + 1u;
        for_member_neighbours(g, F, a, slot, r, [&](int idx27, u32 aj) {
          if (ld_agent(&B.key[aj]) == kbase + (u32)idx27) wmask |= 1u << idx27;
        });
      }
      const u32 cnt = (u32)__popc(wmask);
      u32 v = cnt;
      for (int off = 1; off < 64; off <<= 1) {
        const u32 t = (u32)__shfl_up((int)v, off, 64);
        if (lane >= off) v += t;
      }
      if (lane == 63) s_wave[wave] = v;
      __syncthreads();
      u32 woff = 0u, total = 0u;
      for (int w = 0; w < BFS_T / 64; ++w) {
        if (w < wave) woff += s_wave[w];
        total += s_wave[w];
      }
      const u32 run = s_run;
      u32 pos = base0 + lev_hi + run + woff + (v - cnt);
      if (wmask) {
        u32 m = wmask;
        while (m) {
          const int idx27 = __builtin_ctz(m);
          m &= m - 1u;
          const int dx = idx27 / 9 - 1, dy = (idx27 / 3) % 3 - 1, dz = idx27 % 3 - 1;
          if (pos < base0 + want)
            st_agent(&B.ord[pos], (u32)(a + (long)dx * g.nyz + (long)dy * g.nz + dz));
          else
            B.err[0] = 1u;  // more cells reached than the cluster holds: cannot happen
          ++pos;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_run = run + total;
      __syncthreads();
    }
    const u32 grown = s_run;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (grown == 0u) break;
    lev_lo = lev_hi;
    lev_hi += grown;
    if (lev_hi > want) break;  // (error already flagged)
  }
  if (lev_hi != want && threadIdx.x == 0) B.err[1] = 1u + r;  // the sweep did not reach every cell of the cluster
  // ordered addresses + cluster rank of every cell; the keys go back to "unset" for the next search
  for (u32 k = threadIdx.x; k < want && k < lev_hi; k += BFS_T) {
    const u32 a = ld_agent(&B.ord[base0 + k]);
    B.out_adr[base0 + k] = a;
    B.out_key[base0 + k] = r;
    if (k > 0u || !seedc) B.key[a] = 0xFFFFFFFFu;
  }
}

}  // namespace

void frontier_order_free(fuelmi_frontier* f) {
  OrderScratch* o = f->order;
  if (!o) return;
  void* dev[] = {o->key, o->ord, o->off2, o->err};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (o->h_err) (void)hipHostFree(o->h_err);
  delete o;
  f->order = nullptr;
}

int frontier_reference_order(fuelmi_frontier* f, u32 nq, u32 nkept, u32 n_out, int fin, u32* n_total,
                             std::vector<u32>* h_off2) {
  fuelmi_map* m = f->map;
  FArgs& F = f->F;
  hipStream_t st = f->stream;
  if (!f->order) {
    OrderScratch* o = new OrderScratch;
    f->order = o;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->key), (size_t)m->g.N * sizeof(u32)));
    HIPCHK(hipMemsetAsync(o->key, 0xFF, (size_t)m->g.N * sizeof(u32), st));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->ord), ((size_t)F.cap_q + F.cap_kept) * sizeof(u32)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->off2), ((size_t)F.cap_kept + 1) * sizeof(u32)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o->err), 4 * sizeof(u32)));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&o->h_err), 4 * sizeof(u32), hipHostMallocDefault));
  }
  OrderScratch* o = f->order;
  // cluster r occupies off2[r] .. off2[r+1]: its Q0 cells plus the NQ seed that started it, if one did
  std::vector<u32>& off2 = *h_off2;
  off2.assign(nkept + 1, 0u);
  for (u32 r = 0; r < nkept; ++r) off2[r + 1] = off2[r] + F.h_rec[r].size;  // size counts the seed
  const u32 total = off2[nkept];
  if (total > F.cap_q) {
    fuelmi_set_error("reference order: %u cells exceed the capacity %u", total, F.cap_q);
    return FUELMI_ELIMIT;
  }
  (void)n_out;
  HIPCHK(hipMemcpyAsync(o->off2, off2.data(), (nkept + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
  (void)nq;
  HIPCHK(hipMemsetAsync(o->err, 0, 4 * sizeof(u32), st));
  BArgs B;
  B.key = o->key, B.ord = o->ord, B.off2 = o->off2;
  B.out_adr = F.ms_val[1 - fin], B.out_key = F.ms_key[1 - fin];
  B.nq = nq, B.err = o->err;
  k_bfs_order<<<nkept, BFS_T, 0, st>>>(m->g, F, B);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(o->h_err, o->err, 4 * sizeof(u32), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));  // (off2 is a pageable host vector: its upload has been staged by now)
  if (o->h_err[0] || o->h_err[1]) {
    (void)hipMemsetAsync(o->key, 0xFF, (size_t)m->g.N * sizeof(u32), st);  // (a sweep that stopped half-way leaves keys behind)
    fuelmi_set_error("reference order: the level sweep of cluster %u did not match its cell set", o->h_err[1] - 1u);
    return FUELMI_EHIP;
  }
  *n_total = total;
  return FUELMI_OK;
}
