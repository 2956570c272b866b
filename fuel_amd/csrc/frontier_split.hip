// frontier_split.hip -- FrontierFinder::splitLargeFrontiers / splitHorizontally / downsample /
// computeFrontierInfo (active_perception/src/frontier_finder.cpp:166-242, 374-390, 757-774) on the
// device.
//
// The reference recurses cluster by cluster: down-sample the cells with a pcl::VoxelGrid (leaf =
// down_sample * resolution, leaves aligned to GLOBAL multiples of the leaf size), test whether any
// down-sampled cell lies farther than cluster_size_xy from the mean (xy), and if so cut the cluster
// by the line through the mean perpendicular to the first principal direction of the down-sampled
// cells, then recurse into both halves (dot >= 0 first).  Here all clusters advance one recursion
// LEVEL at a time:
//   k_sp_sums     per cell   : cell count / voxel-index sums of every node evaluated at this level
//                              (block-local LDS table, one global atomic per node and block)
//   k_sp_leaf     per cell   : (node, leaf) -> open-addressing hash table: count + index sums.
//                              The leaf of a voxel is floor(float(centre) * (1/leaf)) exactly like PCL.
//   k_sp_stats    per slot   : centroid; need_split |= dist_xy > cluster_size_xy; covariance sums
//                              (wave-reduced per node, then f64 atomics)
//   k_sp_decide   per node   : FINAL, or SPLIT with the principal direction and two child nodes
//   k_sp_emit     per slot   : centroids of nodes that became FINAL -> filtered-cell output
//   k_sp_part     per cell   : cells of SPLIT nodes move to the child on their side of the cut
// until a level splits nothing (one 64-byte read-back per level).  The leaves of the split forest in
// depth-first order (dot >= 0 half first) are the reference's output order; the cells are regrouped
// by that rank with the same stable multisplit the clustering uses, which keeps them in ascending
// address order, and the per-piece mean / AABB come from the same accumulator kernel.
//
// Arithmetic: node means are exact rationals of integer voxel-index sums evaluated in f64 (the
// reference accumulates f64 cell centres: ~1e-15 apart).  Leaf centroids reproduce pcl::VoxelGrid's
// FLOAT accumulation, over the member cells in ascending voxel address (k_sp_centroid) -- the
// reference feeds PCL the cells in BFS order, so its last float bit can differ; see DESIGN.md.  The
// principal direction uses the closed-form symmetric 2x2 decomposition with the sign convention of
// the oracle (Eigen::EigenSolver is third-party; its sign is unpinned).
#include <cmath>
#include <cstdlib>

#include "frontier_internal.h"

namespace {

enum : u32 { N_ACTIVE = 0, N_SPLIT = 1, N_FINAL = 2, N_DEAD = 3 };

struct SNode {
  u32 orig;    // rank of the region-grown cluster this piece comes from
  u32 path;    // split decisions from the root, 0 = "dot >= 0" half, most recent in bit 0
  u32 depth;   // number of decisions
  u32 level;   // level at which the node is evaluated
  u32 state;
  u32 child0;  // children are child0, child0 + 1
  u32 n;       // cells
  u32 need_split;
  unsigned long long sx, sy, sz;  // voxel-index sums
  unsigned long long nfilt;
  double cov[4];                  // sums of dx*dx, dx*dy, dy*dx, dy*dy over the centroids
  double pc[2];
  u32 rank;  // final rank (FINAL nodes)
  u32 pad;
};

struct LeafAcc {
  u32 cnt;      // member cells
  float c[3];   // centroid (k_sp_centroid)
};

#define SP_EMPTY 0xFFFFFFFFFFFFFFFFull
#define SP_LBITS 14  // signed leaf coordinate bits per axis (+-8192 leaves)
#define SP_LOFF (1 << (SP_LBITS - 1))

struct SArgs {
  u32 n;         // cells (Q0 cells + appended NQ seeds)
  u32* cells;    // voxel addresses
  u32* node;     // current node of every cell
  SNode* nodes;
  u32 cap_nodes;
  u32* ctr;      // [0] n_nodes [1] splits of this level [2] n_filtered [3] overflow [4] n_final [5] used slots
  u64* hkeys;
  LeafAcc* hvals;
  u32* hmemb;    // [table size][memb_cap] voxel addresses of the members of every (node, leaf)
  u32 memb_cap;
  u32* used;     // slots occupied at this level (ctr[5] of them): the per-slot kernels walk this list
  u32 hmask;
  // filtered-cell output
  u32* f_node;
  u64* f_leaf;
  float* f_xyz;
  u32 cap_filt;
  double size_xy;
  float inv_leaf;
  // cfg.reference_order: cells arrive in the reference's BFS order; node means are the reference's sequential
  // f64 sums (computed by the host between levels, nmean[2 * node + {0, 1}]); leaf members are kept by their
  // position in `cells` so that the float centroid accumulates in that order
  int ref;
  const double* nmean;
};

__device__ __forceinline__ void decode(const Geo& g, u32 a, u32& x, u32& y, u32& z) {
  x = a / (u32)g.nyz;
  const u32 r = a - x * (u32)g.nyz;
  y = r / (u32)g.nz;
  z = r - y * (u32)g.nz;
}

__global__ void __launch_bounds__(256)
k_sp_init(SArgs S, const u32* __restrict__ cells_in, const u32* __restrict__ rank_in, u32 n_out,
          const u32* __restrict__ seeds, u32 nseeds, u32 nkept) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_out) {
    S.cells[i] = cells_in[i];
    S.node[i] = rank_in[i];
  } else if (i < n_out + nseeds) {
    S.cells[i] = seeds[2 * (i - n_out)];
    S.node[i] = seeds[2 * (i - n_out) + 1];
  }
  if (i < nkept) {
    SNode nd;
    memset(&nd, 0, sizeof(nd));
    nd.orig = i;
    nd.state = N_ACTIVE;
    S.nodes[i] = nd;
  }
  if (i == 0) {
    S.ctr[0] = nkept;
    S.ctr[1] = S.ctr[2] = S.ctr[3] = S.ctr[4] = S.ctr[5] = 0u;
  }
}

// per-node cell count and index sums for the nodes evaluated at `level`
__global__ void __launch_bounds__(256) k_sp_sums(Geo g, SArgs S, u32 level) {
  __shared__ u32 t_key[128];
  __shared__ u32 t_acc[128][4];
  for (u32 base = blockIdx.x * 1024u; base < S.n; base += gridDim.x * 1024u) {
    if (threadIdx.x < 128) {
      t_key[threadIdx.x] = 0xFFFFFFFFu;
      t_acc[threadIdx.x][0] = t_acc[threadIdx.x][1] = t_acc[threadIdx.x][2] = t_acc[threadIdx.x][3] = 0u;
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
      const u32 i = base + (u32)k * 256u + threadIdx.x;
      if (i >= S.n) continue;
      const u32 nd = S.node[i];
      const SNode& N = S.nodes[nd];
      if (N.state != N_ACTIVE || N.level != level) continue;
      u32 x, y, z;
      decode(g, S.cells[i], x, y, z);
      u32 h = (nd * 2654435761u) >> 25;  // 7 bits
      bool done = false;
      for (int probe = 0; probe < 128 && !done; ++probe) {
        const u32 old = atomicCAS(&t_key[h], 0xFFFFFFFFu, nd);
        if (old == 0xFFFFFFFFu || old == nd) {
          atomicAdd(&t_acc[h][0], 1u);
          atomicAdd(&t_acc[h][1], x);
          atomicAdd(&t_acc[h][2], y);
          atomicAdd(&t_acc[h][3], z);
          done = true;
        }
        h = (h + 1) & 127u;
      }
      if (!done) {  // more than 128 distinct nodes in one 1024-cell chunk: straight to memory
        atomicAdd(&S.nodes[nd].n, 1u);
        atomicAdd(&S.nodes[nd].sx, (unsigned long long)x);
        atomicAdd(&S.nodes[nd].sy, (unsigned long long)y);
        atomicAdd(&S.nodes[nd].sz, (unsigned long long)z);
      }
    }
    __syncthreads();
    if (threadIdx.x < 128 && t_key[threadIdx.x] != 0xFFFFFFFFu) {
      SNode& N = S.nodes[t_key[threadIdx.x]];
      atomicAdd(&N.n, t_acc[threadIdx.x][0]);
      atomicAdd(&N.sx, (unsigned long long)t_acc[threadIdx.x][1]);
      atomicAdd(&N.sy, (unsigned long long)t_acc[threadIdx.x][2]);
      atomicAdd(&N.sz, (unsigned long long)t_acc[threadIdx.x][3]);
    }
    __syncthreads();
  }
}

// between levels only the slots of the previous level need resetting
__global__ void k_sp_clear_used(SArgs S) {
  const u32 n = S.ctr[5];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 h = S.used[i];
    S.hkeys[h] = SP_EMPTY;
    S.hvals[h] = LeafAcc{0u, {0.f, 0.f, 0.f}};
  }
}
__global__ void k_sp_clear(u64* keys, LeafAcc* vals, u32 n) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    keys[i] = SP_EMPTY;
    vals[i] = LeafAcc{0u, {0.f, 0.f, 0.f}};
  }
}

__device__ __forceinline__ u64 sp_mix(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

// (node, leaf) accumulation.  Leaf coordinates as pcl::VoxelGrid computes them: floor(p * inv_leaf)
// in float on the float-converted centre.
__global__ void __launch_bounds__(256) k_sp_leaf(Geo g, SArgs S, u32 level) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < S.n; i += gridDim.x * blockDim.x) {
    const u32 nd = S.node[i];
    const SNode& N = S.nodes[nd];
    if (N.state != N_ACTIVE || N.level != level) continue;
    u32 x, y, z;
    decode(g, S.cells[i], x, y, z);
    const float px = (float)(((double)x + 0.5) * g.res + g.org[0]);
    const float py = (float)(((double)y + 0.5) * g.res + g.org[1]);
    const float pz = (float)(((double)z + 0.5) * g.res + g.org[2]);
    const int lx = (int)floorf(px * S.inv_leaf), ly = (int)floorf(py * S.inv_leaf), lz = (int)floorf(pz * S.inv_leaf);
    const u64 lk = ((u64)(u32)(lz + SP_LOFF) << (2 * SP_LBITS)) | ((u64)(u32)(ly + SP_LOFF) << SP_LBITS) |
                   (u64)(u32)(lx + SP_LOFF);
    const u64 key = ((u64)nd << (3 * SP_LBITS)) | lk;
    u32 h = (u32)sp_mix(key) & S.hmask;
    while (true) {
      const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&S.hkeys[h]), SP_EMPTY, key);
      if (old == SP_EMPTY) S.used[atomicAdd(&S.ctr[5], 1u)] = h;  // first cell of this (node, leaf)
      if (old == SP_EMPTY || old == key) break;
      h = (h + 1) & S.hmask;
    }
    const u32 arr = atomicAdd(&S.hvals[h].cnt, 1u);
    if (arr < S.memb_cap)
      S.hmemb[(size_t)h * S.memb_cap + arr] = S.ref ? i : S.cells[i];
    else
      S.ctr[3] = 1u;
  }
}

// Centroid of every (node, leaf): the float accumulation of pcl::VoxelGrid (centroid += point;
// centroid /= n, all in float) over the member cells in ASCENDING VOXEL ADDRESS -- the order this
// library lists cells in.  Centroids of voxel centres land exactly on voxel faces whenever the members
// are symmetric, so the last float bit decides in which voxel a visibility ray starts: the sum must
// be reproducible, not merely accurate.
__global__ void __launch_bounds__(256) k_sp_centroid(Geo g, SArgs S) {
  const u32 nused = S.ctr[5];
  for (u32 ui = blockIdx.x * blockDim.x + threadIdx.x; ui < nused; ui += gridDim.x * blockDim.x) {
    const u32 h = S.used[ui];
    const u32 n = min(S.hvals[h].cnt, S.memb_cap);
    u32* mb = S.hmemb + (size_t)h * S.memb_cap;
    for (u32 i = 1; i < n; ++i) {  // insertion sort in place (n <= (down_sample + 1)^3)
      const u32 v = mb[i];
      u32 j = i;
      while (j > 0 && mb[j - 1] > v) {
        mb[j] = mb[j - 1];
        --j;
      }
      mb[j] = v;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (u32 i = 0; i < n; ++i) {
      u32 x, y, z;
      decode(g, S.ref ? S.cells[mb[i]] : mb[i], x, y, z);
      sx += (float)(((double)x + 0.5) * g.res + g.org[0]);
      sy += (float)(((double)y + 0.5) * g.res + g.org[1]);
      sz += (float)(((double)z + 0.5) * g.res + g.org[2]);
    }
    const float fn = (float)n;
    S.hvals[h].c[0] = sx / fn;
    S.hvals[h].c[1] = sy / fn;
    S.hvals[h].c[2] = sz / fn;
  }
}

__device__ __forceinline__ void centroid_of(const Geo&, const LeafAcc& a, float c[3]) {
  c[0] = a.c[0], c[1] = a.c[1], c[2] = a.c[2];
}
__device__ __forceinline__ void node_mean_xy(const Geo& g, const SArgs& S, u32 nd, const SNode& N, double m[2]) {
  if (S.ref) {
    m[0] = S.nmean[2 * (size_t)nd];
    m[1] = S.nmean[2 * (size_t)nd + 1];
    return;
  }
  const double n = (double)N.n;
  m[0] = ((double)N.sx / n + 0.5) * g.res + g.org[0];
  m[1] = ((double)N.sy / n + 0.5) * g.res + g.org[1];
}

// need_split and covariance sums of every evaluated node from its centroids.  Table slots are in
// hash order, so a block meets the nodes of the level in random mix: partial sums are collected in a
// block-local LDS table keyed by node (f64 LDS atomics; a wave whose lanes all hold the same node --
// the early levels -- reduces by shuffles first) and flushed with one set of global atomics per node
// and block.
__global__ void __launch_bounds__(256) k_sp_stats(Geo g, SArgs S) {
  __shared__ u32 t_key[256];
  __shared__ double t_cov[256][4];
  __shared__ u32 t_cnt[256];
  __shared__ u32 t_far[256];
  const int lane = threadIdx.x & 63;
  const u32 tsize = S.ctr[5];  // walks the list of occupied slots
  t_far[threadIdx.x] = 0u;
  // few, long-lived blocks: the flush is one set of same-address global atomics per node and block
  t_key[threadIdx.x] = 0xFFFFFFFFu;
  t_cnt[threadIdx.x] = 0u;
  t_cov[threadIdx.x][0] = t_cov[threadIdx.x][1] = t_cov[threadIdx.x][2] = t_cov[threadIdx.x][3] = 0.0;
  __syncthreads();
  for (u32 base = blockIdx.x * 1024u; base < tsize; base += gridDim.x * 1024u) {
    for (int k = 0; k < 4; ++k) {
      const u32 ui = base + (u32)k * 256u + threadIdx.x;
      const u32 h = ui < tsize ? S.used[ui] : 0u;
      const u64 key = ui < tsize ? S.hkeys[h] : SP_EMPTY;
      const bool active = key != SP_EMPTY;
      u32 nd = 0u;
      double dx = 0.0, dy = 0.0;
      u32 far = 0u;
      if (active) {
        nd = (u32)(key >> (3 * SP_LBITS));
        float c[3];
        centroid_of(g, S.hvals[h], c);
        double m[2];
        node_mean_xy(g, S, nd, S.nodes[nd], m);
        dx = (double)c[0] - m[0];
        dy = (double)c[1] - m[1];
        far = sqrt(dx * dx + dy * dy) > S.size_xy ? 1u : 0u;
      }
      const u64 am = __ballot(active);
      if (!am) continue;
      const u32 first = (u32)__shfl((int)nd, __builtin_ctzll(am), 64);
      double v0 = active ? dx * dx : 0.0, v1 = active ? dx * dy : 0.0, v2 = active ? dy * dx : 0.0,
             v3 = active ? dy * dy : 0.0;
      u32 cnt = active ? 1u : 0u;
      bool add = active;
      if (__ballot(active && nd != first) == 0ull) {  // one node in the whole wave
        for (int off = 32; off > 0; off >>= 1) {
          v0 += __shfl_xor(v0, off, 64);
          v1 += __shfl_xor(v1, off, 64);
          v2 += __shfl_xor(v2, off, 64);
          v3 += __shfl_xor(v3, off, 64);
          cnt += (u32)__shfl_xor((int)cnt, off, 64);
        }
        far = __ballot(far != 0u) ? 1u : 0u;
        add = lane == __builtin_ctzll(am);
        nd = first;
      }
      if (add) {
        u32 hh = (nd * 2654435761u) >> 24;  // 8 bits
        bool done = false;
        for (int probe = 0; probe < 256 && !done; ++probe) {
          const u32 old = atomicCAS(&t_key[hh], 0xFFFFFFFFu, nd);
          if (old == 0xFFFFFFFFu || old == nd) {
            atomicAdd(&t_cov[hh][0], v0);
            atomicAdd(&t_cov[hh][1], v1);
            atomicAdd(&t_cov[hh][2], v2);
            atomicAdd(&t_cov[hh][3], v3);
            atomicAdd(&t_cnt[hh], cnt);
            if (far) atomicOr(&t_far[hh], 1u);
            done = true;
          }
          hh = (hh + 1) & 255u;
        }
        if (!done) {  // > 256 distinct nodes in 1024 slots
          SNode& N = S.nodes[nd];
          atomicAdd(&N.cov[0], v0);
          atomicAdd(&N.cov[1], v1);
          atomicAdd(&N.cov[2], v2);
          atomicAdd(&N.cov[3], v3);
          atomicAdd(&N.nfilt, (unsigned long long)cnt);
          if (far) atomicOr(&N.need_split, 1u);
        }
      }
    }
  }
  __syncthreads();
  if (t_key[threadIdx.x] != 0xFFFFFFFFu) {
    SNode& N = S.nodes[t_key[threadIdx.x]];
    atomicAdd(&N.cov[0], t_cov[threadIdx.x][0]);
    atomicAdd(&N.cov[1], t_cov[threadIdx.x][1]);
    atomicAdd(&N.cov[2], t_cov[threadIdx.x][2]);
    atomicAdd(&N.cov[3], t_cov[threadIdx.x][3]);
    atomicAdd(&N.nfilt, (unsigned long long)t_cnt[threadIdx.x]);
    // (atomic, not a plain store: the line also receives memory-side atomics from other XCDs)
    if (t_far[threadIdx.x]) atomicOr(&N.need_split, 1u);
  }
}

// splitHorizontally's decision and principal direction per evaluated node
__global__ void __launch_bounds__(256) k_sp_decide(SArgs S, u32 level, u32 n_nodes) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  SNode& N = S.nodes[i];
  if (N.state != N_ACTIVE || N.level != level) return;
  if (N.n == 0u) {
    N.state = N_DEAD;  // an empty half (the reference would read cells_.front() of an empty vector)
    return;
  }
  if (!N.need_split || N.depth >= 31u) {
    N.state = N_FINAL;
    atomicAdd(&S.ctr[4], 1u);
    return;
  }
  const double nf = (double)N.nfilt;
  const double c00 = N.cov[0] / nf, c01 = N.cov[1] / nf, c10 = N.cov[2] / nf, c11 = N.cov[3] / nf;
  const double a = c00, b = 0.5 * (c01 + c10), d = c11;
  const double tr = a + d, det = a * d - b * b, disc = sqrt(fmax(tr * tr / 4 - det, 0.0));
  const double l = tr / 2 + disc;
  double vx = b, vy = l - a;
  if (fabs(vx) + fabs(vy) < 1e-300) {
    vx = l - d;
    vy = b;
  }
  if (fabs(vx) + fabs(vy) < 1e-300) {
    vx = 1;
    vy = 0;
  }
  const double nn = sqrt(vx * vx + vy * vy);
  N.pc[0] = vx / nn;
  N.pc[1] = vy / nn;
  const u32 c0 = atomicAdd(&S.ctr[0], 2u);
  if (c0 + 2u > S.cap_nodes) {
    S.ctr[3] = 1u;
    N.state = N_FINAL;
    atomicAdd(&S.ctr[4], 1u);
    return;
  }
  for (u32 k = 0; k < 2u; ++k) {
    SNode ch;
    memset(&ch, 0, sizeof(ch));
    ch.orig = N.orig;
    ch.path = (N.path << 1) | k;
    ch.depth = N.depth + 1u;
    ch.level = level + 1u;
    ch.state = N_ACTIVE;
    S.nodes[c0 + k] = ch;
  }
  N.child0 = c0;
  N.state = N_SPLIT;
  atomicAdd(&S.ctr[1], 1u);
}

// centroids of the nodes that became FINAL at this level = their filtered_cells_
__global__ void __launch_bounds__(256) k_sp_emit(Geo g, SArgs S, u32 level) {
  const u32 nused = S.ctr[5];
  for (u32 ui = blockIdx.x * blockDim.x + threadIdx.x; ui < nused; ui += gridDim.x * blockDim.x) {
    const u32 h = S.used[ui];
    const u64 key = S.hkeys[h];
    const u32 nd = (u32)(key >> (3 * SP_LBITS));
    const SNode& N = S.nodes[nd];
    if (N.state != N_FINAL || N.level != level) continue;
    const u32 o = atomicAdd(&S.ctr[2], 1u);
    if (o >= S.cap_filt) {
      S.ctr[3] = 1u;
      continue;
    }
    float c[3];
    centroid_of(g, S.hvals[h], c);
    S.f_node[o] = nd;
    S.f_leaf[o] = key & ((1ull << (3 * SP_LBITS)) - 1ull);  // (lz, ly, lx): PCL's output order
    S.f_xyz[3 * o] = c[0], S.f_xyz[3 * o + 1] = c[1], S.f_xyz[3 * o + 2] = c[2];
  }
}

// cells of SPLIT nodes go to the child on their side: (cell_xy - mean_xy) . pc >= 0 -> first child
__global__ void __launch_bounds__(256) k_sp_part(Geo g, SArgs S, u32 level) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < S.n; i += gridDim.x * blockDim.x) {
    const u32 nd = S.node[i];
    const SNode& N = S.nodes[nd];
    if (N.state != N_SPLIT || N.level != level) continue;
    u32 x, y, z;
    decode(g, S.cells[i], x, y, z);
    double m[2];
    node_mean_xy(g, S, nd, N, m);
    const double px = ((double)x + 0.5) * g.res + g.org[0], py = ((double)y + 0.5) * g.res + g.org[1];
    const double dot = (px - m[0]) * N.pc[0] + (py - m[1]) * N.pc[1];
    S.node[i] = N.child0 + (dot >= 0 ? 0u : 1u);
  }
}

__global__ void k_sp_next_level(SArgs S) {  // after k_sp_clear_used
  if (threadIdx.x == 0 && blockIdx.x == 0) S.ctr[1] = S.ctr[5] = 0u;
}

// depth-first order of the FINAL nodes: key = (orig, path left-aligned); rank + cell offsets, and
// the cluster records the regrouping stage expects
__global__ void __launch_bounds__(256) k_sp_rank(SArgs S, u32 n_nodes, KeptRec* krec, u32* counts) {
  __shared__ u64 t_key[256];
  __shared__ u32 t_n[256];
  auto key_of = [](const SNode& M) {
    const u32 pl = M.depth ? (M.path << (32u - M.depth)) : 0u;
    return ((u64)M.orig << 32) | pl;
  };
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool mine = i < n_nodes && S.nodes[i].state == N_FINAL;
  const u64 ki = mine ? key_of(S.nodes[i]) : 0ull;
  u32 rank = 0u, off = 0u;
  for (u32 base = 0; base < n_nodes; base += 256u) {  // all nodes, one LDS tile at a time
    const u32 j = base + threadIdx.x;
    const bool fin = j < n_nodes && S.nodes[j].state == N_FINAL;
    t_key[threadIdx.x] = fin ? key_of(S.nodes[j]) : ~0ull;
    t_n[threadIdx.x] = fin ? S.nodes[j].n : 0u;
    __syncthreads();
    if (mine)
      for (int t = 0; t < 256; ++t)
        if (t_key[t] < ki) {
          ++rank;
          off += t_n[t];
        }
    __syncthreads();
  }
  if (!mine) return;
  SNode& N = S.nodes[i];
  N.rank = rank;
  KeptRec& r = krec[rank];
  r.addr = 0u, r.slot = 0u, r.size = N.n, r.off = off;
  r.sum[0] = r.sum[1] = r.sum[2] = 0ull;
  for (int k = 0; k < 3; ++k) r.box[k] = 0xFFFFFFFFu, r.box[3 + k] = 0u;
  if (rank == S.ctr[4] - 1u) {
    counts[0] = S.n;
    counts[1] = 0u;
    counts[2] = 0u;
    counts[3] = S.ctr[4];
    counts[5] = S.n;
  }
}

__global__ void __launch_bounds__(256) k_sp_keys(SArgs S, u32* key_out, u32* val_out) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < S.n; i += gridDim.x * blockDim.x) {
    const SNode& N = S.nodes[S.node[i]];
    key_out[i] = N.state == N_FINAL ? N.rank : 0xFFFFFFFFu;
    val_out[i] = S.cells[i];
  }
}
__global__ void __launch_bounds__(256) k_sp_filt_rank(SArgs S, u32 nfilt) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nfilt; i += gridDim.x * blockDim.x)
    S.f_node[i] = S.nodes[S.f_node[i]].rank;
}

}  // namespace

struct SplitScratch {
  u32 cap_cells = 0, cap_nodes = 0, cap_filt = 0, tsize = 0;
  u32 *cells = nullptr, *node = nullptr, *ctr = nullptr, *seeds = nullptr, *counts = nullptr;
  SNode* nodes = nullptr;
  u64* hkeys = nullptr;
  LeafAcc* hvals = nullptr;
  u32* hmemb = nullptr;
  u32* used = nullptr;
  u32 memb_cap = 0;
  u32* f_node = nullptr;
  u64* f_leaf = nullptr;
  float* f_xyz = nullptr;
  u32* h_ctr = nullptr;  // pinned [16]
  double* nmean = nullptr;  // [cap_nodes][2] (reference order)
};

void frontier_split_free(fuelmi_frontier* f) {
  SplitScratch* s = f->split;
  if (!s) return;
  void* dev[] = {s->cells, s->node,  s->ctr,    s->seeds,  s->counts, s->nodes,  s->nmean,
                 s->hkeys, s->hvals, s->hmemb, s->used,   s->f_node, s->f_leaf, s->f_xyz};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (s->h_ctr) (void)hipHostFree(s->h_ctr);
  delete s;
  f->split = nullptr;
}

template <typename T>
static int sp_alloc(T** p, size_t n) {
  HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return FUELMI_OK;
}

static int split_ensure(fuelmi_frontier* f, u32 n) {
  if (!f->split) f->split = new SplitScratch;
  SplitScratch* s = f->split;
  int rc;
  if (!s->ctr) {
    s->cap_nodes = 2u * f->F.cap_kept;
    if ((rc = sp_alloc(&s->ctr, 16)) || (rc = sp_alloc(&s->counts, 16)) || (rc = sp_alloc(&s->nodes, s->cap_nodes)) ||
        (rc = sp_alloc(&s->nmean, 2 * (size_t)s->cap_nodes)) ||
        (rc = sp_alloc(&s->seeds, 2 * (size_t)f->F.cap_kept)))
      return rc;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&s->h_ctr), 64, hipHostMallocDefault));
  }
  const u32 ds1 = (u32)f->cfg.down_sample + 1u;  // a leaf holds down_sample^3 voxel centres; +1 per axis of slack
  if (s->memb_cap != ds1 * ds1 * ds1) {
    s->memb_cap = ds1 * ds1 * ds1;
    s->cap_cells = 0;  // force (re)allocation of the table with the new member capacity
  }
  if (n > s->cap_cells) {
    void* old[] = {s->cells, s->node, s->hkeys, s->hvals, s->hmemb, s->used, s->f_node, s->f_leaf, s->f_xyz};
    for (void* p : old)
      if (p) HIPCHK(hipFree(p));
    s->cells = s->node = s->f_node = nullptr;
    s->hkeys = s->f_leaf = nullptr;
    s->hvals = nullptr;
    s->hmemb = nullptr;
    s->used = nullptr;
    s->f_xyz = nullptr;
    s->cap_cells = 0;
    const u32 cap = n + n / 4 + 4096;
    u32 t = 1024;
    while (t < 2u * cap) t <<= 1;
    if ((rc = sp_alloc(&s->cells, cap)) || (rc = sp_alloc(&s->node, cap)) || (rc = sp_alloc(&s->hkeys, t)) ||
        (rc = sp_alloc(&s->hvals, t)) || (rc = sp_alloc(&s->hmemb, (size_t)t * s->memb_cap)) ||
        (rc = sp_alloc(&s->used, cap)) || (rc = sp_alloc(&s->f_node, cap)) || (rc = sp_alloc(&s->f_leaf, cap)) ||
        (rc = sp_alloc(&s->f_xyz, 3 * (size_t)cap)))
      return rc;
    s->cap_cells = cap;
    s->cap_filt = cap;
    s->tsize = t;
  }
  return FUELMI_OK;
}

int frontier_split_run(fuelmi_frontier* f, u32 nq, u32 nkept, u32 n_out, int fin, u32* n_final, u32* n_cells,
                       std::vector<std::vector<float>>* filtered) {
  fuelmi_map* m = f->map;
  const Geo& g = m->g;
  FArgs& F = f->F;
  hipStream_t st = f->stream;
  if (f->cfg.down_sample <= 0 || f->cfg.down_sample > 6 || !(f->cfg.cluster_size_xy > 0.0)) {
    fuelmi_set_error("frontier split needs 0 < down_sample <= 6 and cluster_size_xy > 0");
    return FUELMI_EINVAL;
  }
  if (nkept >= (1u << (64 - 3 * SP_LBITS - 1))) {
    fuelmi_set_error("frontier split: too many clusters (%u)", nkept);
    return FUELMI_ELIMIT;
  }
  // NQ seeds that started a kept cluster are not in the grouped Q0 array: append them (in reference order
  // they already lead their cluster in the input)
  const bool ref = f->ref_now;  // (the order this search's cells arrive in: cfg.reference_order, resolved per search)
  std::vector<u32> seeds;
  for (u32 r = 0; r < nkept && !ref; ++r)
    if (F.h_rec[r].slot >= nq) {
      seeds.push_back(F.h_rec[r].addr);
      seeds.push_back(r);
    }
  const u32 nseeds = (u32)(seeds.size() / 2);
  const u32 n = n_out + nseeds;
  if (n > F.cap_q) {
    fuelmi_set_error("frontier split: %u cells exceed the capacity %u", n, F.cap_q);
    return FUELMI_ELIMIT;
  }
  int rc = split_ensure(f, n);
  if (rc) return rc;
  SplitScratch* s = f->split;
  if (nseeds) HIPCHK(hipMemcpyAsync(s->seeds, seeds.data(), seeds.size() * sizeof(u32), hipMemcpyHostToDevice, st));

  SArgs S;
  S.n = n;
  S.cells = s->cells, S.node = s->node, S.nodes = s->nodes, S.cap_nodes = s->cap_nodes, S.ctr = s->ctr;
  S.hkeys = s->hkeys, S.hvals = s->hvals, S.hmemb = s->hmemb, S.memb_cap = s->memb_cap, S.used = s->used;
  u32 t = 1024;  // table sized to this search (cleared every level)
  while (t < 2u * n) t <<= 1;
  S.hmask = t - 1u;
  S.f_node = s->f_node, S.f_leaf = s->f_leaf, S.f_xyz = s->f_xyz, S.cap_filt = s->cap_filt;
  S.size_xy = f->cfg.cluster_size_xy;
  const float leaf = (float)(g.res * (double)f->cfg.down_sample);  // setLeafSize(float...)
  S.inv_leaf = 1.0f / leaf;
  S.ref = ref ? 1 : 0;
  S.nmean = s->nmean;
  // reference order: host copies of the cells (once) and of their node labels (every level) for the
  // sequential means; the nodes evaluated at one level are a contiguous id range (children are allocated in
  // pairs from one counter)
  std::vector<u32> h_cells_ord, h_node;
  std::vector<double> h_mean;
  u32 lvl_lo = 0u, lvl_hi = nkept;
  if (ref) {
    h_cells_ord.resize(n);
    h_node.resize(n);
    HIPCHK(hipMemcpyAsync(h_cells_ord.data(), F.ms_val[fin], (size_t)n * sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_node.data(), F.ms_key[fin], (size_t)n * sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  // average_ of every node in [lo, hi) as computeFrontierInfo sums it (:374-390): cell centres added one by one
  // in list order, then divided by the count
  auto seq_means = [&](u32 lo, u32 hi) -> int {
    const u32 cnt = hi - lo;
    if (cnt == 0u) return FUELMI_OK;
    std::vector<double> sum(3 * (size_t)cnt, 0.0);
    std::vector<u32> num(cnt, 0u);
    for (u32 i = 0; i < n; ++i) {
      const u32 nd = h_node[i];
      if (nd < lo || nd >= hi) continue;
      const u32 a = h_cells_ord[i];
      const u32 x = a / (u32)g.nyz, rr = a - x * (u32)g.nyz, y = rr / (u32)g.nz, z = rr - y * (u32)g.nz;
      double* sm = &sum[3 * (size_t)(nd - lo)];
      sm[0] += ((double)x + 0.5) * g.res + g.org[0];
      sm[1] += ((double)y + 0.5) * g.res + g.org[1];
      sm[2] += ((double)z + 0.5) * g.res + g.org[2];
      ++num[nd - lo];
    }
    h_mean.resize(2 * (size_t)cnt);
    for (u32 k = 0; k < cnt; ++k) {
      const double dn = (double)num[k];
      h_mean[2 * (size_t)k] = num[k] ? sum[3 * (size_t)k] / dn : 0.0;
      h_mean[2 * (size_t)k + 1] = num[k] ? sum[3 * (size_t)k + 1] / dn : 0.0;
    }
    HIPCHK(hipMemcpyAsync(s->nmean + 2 * (size_t)lo, h_mean.data(), 2 * (size_t)cnt * sizeof(double),
                          hipMemcpyHostToDevice, st));
    return FUELMI_OK;
  };

  const int gb = (int)std::min<u32>(2048u, (n + 255u) / 256u);
  const int gb4 = (int)std::min<u32>(2048u, (n + 1023u) / 1024u);
  const int gt = (int)std::min<u32>(2048u, (t + 255u) / 256u);  // full-table clear (once)
  k_sp_init<<<(std::max(n, nkept) + 255) / 256, 256, 0, st>>>(S, F.ms_val[fin], F.ms_key[fin], n_out, s->seeds, nseeds,
                                                                nkept);
  u32 n_nodes = nkept;
  k_sp_clear<<<gt, 256, 0, st>>>(S.hkeys, S.hvals, t);
  for (u32 level = 0; level < 40u; ++level) {
    if (ref && (rc = seq_means(lvl_lo, lvl_hi))) return rc;
    k_sp_sums<<<gb4, 256, 0, st>>>(g, S, level);
    k_sp_leaf<<<gb, 256, 0, st>>>(g, S, level);
    k_sp_centroid<<<256, 256, 0, st>>>(g, S);
    k_sp_stats<<<128, 256, 0, st>>>(g, S);
    k_sp_decide<<<(n_nodes + 255) / 256, 256, 0, st>>>(S, level, n_nodes);
    k_sp_emit<<<256, 256, 0, st>>>(g, S, level);
    k_sp_part<<<gb, 256, 0, st>>>(g, S, level);
    HIPCHK(hipMemcpyAsync(s->h_ctr, s->ctr, 16 * sizeof(u32), hipMemcpyDeviceToHost, st));
    if (ref) HIPCHK(hipMemcpyAsync(h_node.data(), S.node, (size_t)n * sizeof(u32), hipMemcpyDeviceToHost, st));
    k_sp_clear_used<<<256, 256, 0, st>>>(S);
    k_sp_next_level<<<1, 64, 0, st>>>(S);
    HIPCHK(hipStreamSynchronize(st));  // (h_mean's upload has completed as well: the vector is reused)
    if (s->h_ctr[3]) {
      fuelmi_set_error("frontier split capacity exceeded (nodes %u/%u, filtered %u/%u)", s->h_ctr[0], s->cap_nodes,
                       s->h_ctr[2], s->cap_filt);
      return FUELMI_ELIMIT;
    }
    lvl_lo = n_nodes;  // the children created at this level are evaluated at the next
    n_nodes = s->h_ctr[0];
    lvl_hi = n_nodes;
    if (s->h_ctr[1] == 0u) break;
  }
  const u32 nfinal = s->h_ctr[4], nfilt = s->h_ctr[2];
  if (nfinal == 0u || nfinal > F.cap_kept) {
    fuelmi_set_error("frontier split produced %u clusters (capacity %u)", nfinal, F.cap_kept);
    return FUELMI_ELIMIT;
  }

  // regroup the cells by depth-first rank and refill the result buffers
  FArgs F2 = F;
  F2.counts = s->counts;
  F2.keys_from_slots = 0;  // keys/values come from k_sp_keys
  HIPCHK(hipMemsetAsync(s->counts, 0, 16 * sizeof(u32), st));
  k_sp_rank<<<(n_nodes + 255) / 256, 256, 0, st>>>(S, n_nodes, F.krec, s->counts);
  k_sp_keys<<<gb, 256, 0, st>>>(S, F.ms_key[0], F.ms_val[0]);
  if (nfilt) k_sp_filt_rank<<<(nfilt + 255) / 256, 256, 0, st>>>(S, nfilt);
  HIPCHK(hipGetLastError());
  rc = frontier_regroup(f, F2, nfinal > 256 ? 2 : 1);
  if (rc) return rc;
  std::vector<u32> fn(nfilt);
  std::vector<u64> fl(nfilt);
  std::vector<float> fx(3 * (size_t)nfilt);
  if (nfilt) {
    HIPCHK(hipMemcpyAsync(fn.data(), s->f_node, nfilt * sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(fl.data(), s->f_leaf, nfilt * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(fx.data(), s->f_xyz, 3 * (size_t)nfilt * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));

  // filtered cells per piece, ascending leaf index (z-major like PCL's idx = i + j*dx + k*dx*dy):
  // bucket by piece, then order the ~100 entries of each bucket
  std::vector<u32> start(nfinal + 1, 0u);
  for (u32 i = 0; i < nfilt; ++i)
    if (fn[i] < nfinal) ++start[fn[i] + 1];
  for (u32 r = 0; r < nfinal; ++r) start[r + 1] += start[r];
  std::vector<u32> order(start[nfinal]), fill(start.begin(), start.end() - 1);
  for (u32 i = 0; i < nfilt; ++i)
    if (fn[i] < nfinal) order[fill[fn[i]]++] = i;
  filtered->assign(nfinal, std::vector<float>());
  for (u32 r = 0; r < nfinal; ++r) {
    std::sort(order.begin() + start[r], order.begin() + start[r + 1], [&](u32 a, u32 b) { return fl[a] < fl[b]; });
    std::vector<float>& v = (*filtered)[r];
    v.reserve(3 * (size_t)(start[r + 1] - start[r]));
    for (u32 k = start[r]; k < start[r + 1]; ++k) {
      const u32 i = order[k];
      v.push_back(fx[3 * (size_t)i]);
      v.push_back(fx[3 * (size_t)i + 1]);
      v.push_back(fx[3 * (size_t)i + 2]);
    }
  }
  *n_final = nfinal;
  *n_cells = n;
  return FUELMI_OK;
}
