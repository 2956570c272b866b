// esdf.hip -- box-local exact Euclidean distance transform (SDFMap::updateESDF3d,
// plan_env/src/sdf_map.cpp:152-241; 1-D pass = fillESDF :116-150).
//
// The reference computes, inside [local_bound_min_, local_bound_max_], three 1-D lower-envelope
// passes (z, y, x) over exact-integer squared distances held in doubles, then res*sqrt().  The
// passes compute D(q) = min_p f(p) + (q-p)^2 exactly (all integers < 2^24), so any exact
// evaluation of that min-plus gives bit-identical squared distances.  On the GPU:
//   * z pass  : sources come from bit-planes; the distance to the nearest set bit of a z-line is
//               two ctz/clz scans -- no sequential envelope.  Fused into the y-pass kernel: the
//               WG fills an LDS tile f[y][z-chunk] = dz^2 (u16) straight from the planes.
//   * y, x    : every output voxel is a lane; it scans outward r = 1,2,.. over the LDS-staged
//               line while r^2 < best ("pruned brute force").  Work is O(distance) per voxel,
//               fully parallel, no per-line sequential stack.  Source voxels cost nothing.
// HBM traffic per box voxel: planes 2 x 1/8 B read, y-pass result u32 written + read, f32
// distance written = 12.25 B.
#include <algorithm>
#include <cmath>
#include <vector>

#include "fuelmi_internal.h"

// MODE 0: sources = inflated | unknown (optimistic_ == false, sdf_map.cpp:169-182)
// MODE 1: sources = inflated            (optimistic_ == true,  :156-167)
// MODE 2: sources = !inflated           (signed_dist_ negative pass, :203-215)
template <int MODE>
__device__ __forceinline__ u64 src_word(const u64* __restrict__ infl, const u64* __restrict__ unk, long w) {
  if (MODE == 0) return infl[w] | unk[w];
  if (MODE == 1) return infl[w];
  return ~infl[w];
}

// distance (in voxels) from z to the nearest source of the line inside [zlo, zhi]; -1 if none
template <int MODE>
__device__ __forceinline__ int nearest_src(const u64* __restrict__ infl, const u64* __restrict__ unk,
                                           long linebit, int zlo, int zhi, int z) {
  int best = -1;
  {
    long pos = linebit + z, end = linebit + zhi;
    long w = pos >> 6;
    u64 word = src_word<MODE>(infl, unk, w) & (~0ull << (pos & 63));
    while (true) {
      if (word) {
        long c = (w << 6) + __builtin_ctzll(word);
        if (c <= end) best = (int)(c - pos);
        break;
      }
      ++w;
      if ((w << 6) > end) break;
      word = src_word<MODE>(infl, unk, w);
    }
  }
  if (best != 0) {
    long pos = linebit + z - 1, beg = linebit + zlo;
    if (pos >= beg) {
      long w = pos >> 6;
      u64 word = src_word<MODE>(infl, unk, w) & (~0ull >> (63 - (int)(pos & 63)));
      while (true) {
        if (word) {
          long c = (w << 6) + 63 - __builtin_clzll(word);
          if (c >= beg) {
            int d = (int)(linebit + z - c);
            if (best < 0 || d < best) best = d;
          }
          break;
        }
        --w;
        if ((w << 6) + 63 < beg) break;
        word = src_word<MODE>(infl, unk, w);
      }
    }
  }
  return best;
}

// fused z + y pass.  grid.x = (#x in box) * nzc ; block handles one x and one z-chunk of ZC.
template <int MODE>
__global__ void __launch_bounds__(256)
k_esdf_zy(Geo g, Box3 b, const u64* __restrict__ infl, const u64* __restrict__ unk, u32* __restrict__ tmp,
          int ZC, int nzc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* tile = reinterpret_cast<unsigned short*>(smem_raw);  // [ylen][ZC] dz^2 or INF16
  const int x = b.lo[0] + blockIdx.x / nzc;
  const int zc = blockIdx.x % nzc;
  const int z0 = b.lo[2] + zc * ZC;
  const int zcnt = min(ZC, b.hi[2] - z0 + 1);
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int total = ylen * ZC;
  const int T = blockDim.x;
  const int dy = T / ZC, dz = T - dy * ZC;  // incremental (yi, zi) stepping, no division in loops

  {
    int yi = threadIdx.x / ZC, zi = threadIdx.x - yi * ZC;
    for (int o = threadIdx.x; o < total; o += T) {
      unsigned short v = INF16;
      if (zi < zcnt) {
        long linebit = (long)x * g.nyz + (long)(b.lo[1] + yi) * g.nz;
        int d = nearest_src<MODE>(infl, unk, linebit, b.lo[2], b.hi[2], z0 + zi);
        if (d >= 0) v = (unsigned short)(d * d);
      }
      tile[o] = v;
      yi += dy;
      zi += dz;
      if (zi >= ZC) {
        zi -= ZC;
        ++yi;
      }
    }
  }
  __syncthreads();
  {
    int yi = threadIdx.x / ZC, zi = threadIdx.x - yi * ZC;
    for (int o = threadIdx.x; o < total; o += T) {
      if (zi < zcnt) {
        unsigned short v0 = tile[o];
        u32 best = (v0 == INF16) ? INF32 : (u32)v0;
        if (best != 0u) {
          const int rmax = max(yi, ylen - 1 - yi);
          for (int r = 1; r <= rmax && (u32)(r * r) < best; ++r) {
            const u32 rr = (u32)(r * r);
            if (yi - r >= 0) {
              unsigned short v = tile[o - r * ZC];
              if (v != INF16) best = min(best, (u32)v + rr);
            }
            if (yi + r < ylen) {
              unsigned short v = tile[o + r * ZC];
              if (v != INF16) best = min(best, (u32)v + rr);
            }
          }
        }
        tmp[(long)x * g.nyz + (long)(b.lo[1] + yi) * g.nz + z0 + zi] = best;
      }
      yi += dy;
      zi += dz;
      if (zi >= ZC) {
        zi -= ZC;
        ++yi;
      }
    }
  }
}

// distance_buffer_ value of a squared voxel distance.  f32 arithmetic: best < 2^24 is exact in f32,
// sqrtf is correctly rounded, so the result is within ~1.2 ulp (< 5e-6 m at 40 m) of the reference's
// f64 res*sqrt(D) -- the f64 sqrt sequence made the x pass VALU-bound.
// v_sqrt_f32 directly (1 ulp: <= 1e-5 m at 100 m, the parity bar is 1e-4): the correctly rounded sqrtf
// expands to ~15 instructions per value, more than the envelope scan of a typical voxel
__device__ __forceinline__ float esdf_sqrt(u32 best) { return __builtin_amdgcn_sqrtf((float)best); }
__device__ __forceinline__ float esdf_out(u32 best, float res) {
  return (best >= INF32) ? INFINITY : res * esdf_sqrt(best);
}
__device__ __forceinline__ float esdf_merge_neg(float cur, u32 best, float res) {
  if (best >= INF32) return -INFINITY;  // reference: += -(res*sqrt(DBL_MAX)) + res
  if (best == 0u) return cur;
  return cur - res * esdf_sqrt(best) + res;
}

// x pass.  The (y,z) columns of the box are enumerated j = yy*zlen + zz; a block stages S
// consecutive columns for every x of the box in LDS (rows of S*4 B are contiguous in memory
// whenever the box spans full z-lines) and every lane scans its own column.
// OUT 0: distance_buffer_ = res*sqrt(D)                      (sdf_map.cpp:191-199)
// OUT 1: negative pass merged in place                       (sdf_map.cpp:216-240):
//        dneg = res*sqrt(D); if (dneg > 0) distance_buffer_ += -dneg + res
template <int S, int OUT>
__global__ void __launch_bounds__(256)
k_esdf_x(Geo g, Box3 b, const u32* __restrict__ tmp, float* __restrict__ dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32* tile = reinterpret_cast<u32*>(smem_raw);  // [xlen][S]
  const int xlen = b.hi[0] - b.lo[0] + 1;
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int zlen = b.hi[2] - b.lo[2] + 1;
  const int ncol = ylen * zlen;
  const int c = threadIdx.x % S;
  const int col = blockIdx.x * S + c;
  const bool valid = col < ncol;
  const int yy = valid ? col / zlen : 0;
  const int zz = valid ? col - yy * zlen : 0;
  const long coloff = (long)(b.lo[1] + yy) * g.nz + b.lo[2] + zz;
  const int rows = blockDim.x / S;
  const float resf = (float)g.res;
  for (int xi = threadIdx.x / S; xi < xlen; xi += rows)
    tile[xi * S + c] = valid ? tmp[(long)(b.lo[0] + xi) * g.nyz + coloff] : INF32;
  __syncthreads();
  if (!valid) return;
  for (int xi = threadIdx.x / S; xi < xlen; xi += rows) {
    u32 best = tile[xi * S + c];
    if (best != 0u) {
      const int rmax = max(xi, xlen - 1 - xi);
      for (int r = 1; r <= rmax && (u32)(r * r) < best; ++r) {
        const u32 rr = (u32)(r * r);
        if (xi - r >= 0) best = min(best, tile[(xi - r) * S + c] + rr);
        if (xi + r < xlen) best = min(best, tile[(xi + r) * S + c] + rr);
      }
    }
    const long a = (long)(b.lo[0] + xi) * g.nyz + coloff;
    if (OUT == 0)
      dist[a] = esdf_out(best, resf);
    else
      dist[a] = esdf_merge_neg(dist[a], best, resf);
  }
}

// ------------------------------------------------------------------------------------------------
// Vectorised variants (nz % 4 == 0): every lane owns 4 z-adjacent voxels, so LDS traffic is
// ds_read_b64 / ds_read_b128, global traffic is 16 B per lane, and the 4 outward scans of a lane
// share one loop (z-neighbours have near-equal distances: the field is 1-Lipschitz).
// The z range is processed on its 4-aligned superset [z0a, z1a]; voxels outside the true box get
// f = INF and are never written.
// ------------------------------------------------------------------------------------------------
// highest / lowest source position of a line inside [zlo, zhi], or -1
template <int MODE>
__device__ __forceinline__ int line_src_down(const u64* __restrict__ infl, const u64* __restrict__ unk,
                                             long linebit, int zlo, int zhi) {
  if (zhi < zlo) return -1;
  long pos = linebit + zhi, beg = linebit + zlo;
  long w = pos >> 6;
  u64 word = src_word<MODE>(infl, unk, w) & (~0ull >> (63 - (int)(pos & 63)));
  while (true) {
    if (word) {
      long c = (w << 6) + 63 - __builtin_clzll(word);
      return c >= beg ? (int)(c - linebit) : -1;
    }
    --w;
    if ((w << 6) + 63 < beg) return -1;
    word = src_word<MODE>(infl, unk, w);
  }
}
template <int MODE>
__device__ __forceinline__ int line_src_up(const u64* __restrict__ infl, const u64* __restrict__ unk,
                                           long linebit, int zlo, int zhi) {
  if (zhi < zlo) return -1;
  long pos = linebit + zlo, end = linebit + zhi;
  long w = pos >> 6;
  u64 word = src_word<MODE>(infl, unk, w) & (~0ull << (pos & 63));
  while (true) {
    if (word) {
      long c = (w << 6) + __builtin_ctzll(word);
      return c <= end ? (int)(c - linebit) : -1;
    }
    ++w;
    if ((w << 6) > end) return -1;
    word = src_word<MODE>(infl, unk, w);
  }
}

// ------------------------------------------------------------------------------------------------
// Outward scan of one line for 4 z-adjacent outputs.
// The plain scan looks at rows i +- r while r^2 < best: O(distance) LDS reads per output -- ~2 in a half-explored
// map (unknown voxels are sources), ~100 in an explored hall whose only source is the floor.  The FAR kernels
// (picked by the host when most outputs of the previous update were far from sources, esdf_use_far) add two
// exact lower bounds over the staged tile:
//   cm[col]    = min f over the whole line: no candidate f(p) + r^2, r >= 1, beats best when cm + 1 >= best
//                (flat fields -- most lines of an empty hall -- end here, with one LDS read);
//   bm[k][col] = min f over rows 8k .. 8k+7: after `near` rows each side the scan walks BLOCKS outwards, and a
//                block whose nearest row is d away is skipped when bm + d^2 >= best (all of its candidates are
//                at least that).
// Both bounds only remove candidates that cannot lower the minimum: D(q) = min_p f(p) + (q-p)^2 stays exact.
// Building them costs a pass over the tile and a barrier (~5 us per kernel on the 400^2 x 100 map), which is why
// the half-explored regime keeps the plain kernels.
// ------------------------------------------------------------------------------------------------
#define ESDF_FAR_D 8u  // statistic: an output counts as "far from sources" beyond this many voxels
#define ESDF_NG 256    // groups of 16 x-slabs the statistic is kept for (maps wider than 4096 voxels share groups)
// rows of the fused z/y pass that need no sweep (all sources / no source) take a short cut: measured 38.8 -> 37.2 us on
// 400^2 x 100, 188 -> 172 us on 800^2 x 200
static bool zy_fastrow() { return true; }
static int esdf_near() { return 2; }  // rows the FAR kernels scan one by one before the block walk (even: two rows per trip)
// one 16-byte global store that stays one instruction (a plain uint4 assignment next to the per-component edge
// path came out as a 12-byte plus a 4-byte store)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* p, uint4 v) {
  u32x4_t w = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<u32x4_t*>(p) = w;
}
__device__ __forceinline__ uint4 lds4(const unsigned char* p, int off) { return *reinterpret_cast<const uint4*>(p + off); }
__device__ __forceinline__ u32 max4(u32 a, u32 b, u32 c, u32 d) { return max(max(a, b), max(c, d)); }

// block and line minima of a staged tile: rows of `stride` bytes, `vec_per_row` uint4 per row, n rows.  cm must
// hold INF32 before the call (with a barrier in between); a barrier must follow before scan_far4.
__device__ __forceinline__ void build_block_minima(const unsigned char* tile, unsigned char* bm, u32* cm, int stride,
                                                   int vec_per_row, int n) {
  const int nblk = (n + 7) >> 3;
  for (int it = threadIdx.x; it < nblk * vec_per_row; it += blockDim.x) {
    const int k = it / vec_per_row, v = it - k * vec_per_row;
    const int last = (n - 1) * stride + 16 * v;
    int o = __mul24(8 * k, stride) + 16 * v;
    uint4 m = lds4(tile, o);
#pragma unroll
    for (int t = 1; t < 8; ++t) {
      o = min(o + stride, last);
      const uint4 q = lds4(tile, o);
      m.x = min(m.x, q.x), m.y = min(m.y, q.y), m.z = min(m.z, q.z), m.w = min(m.w, q.w);
    }
    *reinterpret_cast<uint4*>(bm + __mul24(k, stride) + 16 * v) = m;
    atomicMin(cm + 4 * v, m.x);
    atomicMin(cm + 4 * v + 1, m.y);
    atomicMin(cm + 4 * v + 2, m.z);
    atomicMin(cm + 4 * v + 3, m.w);
  }
}

// near phase for the 4 columns at byte offset `col` of row i: min over |p - i| <= near (all p when near == 0).
// open = true when rows further out can still lower one of the four minima.
__device__ __forceinline__ uint4 scan_near4(const unsigned char* tile, int stride, int n, int i, int col, int near,
                                            bool& open) {
  const int base = __mul24(i, stride) + col;
  const uint4 v0 = lds4(tile, base);
  u32 b0 = v0.x, b1 = v0.y, b2 = v0.z, b3 = v0.w;
  u32 mx = max4(b0, b1, b2, b3);
  const u32 cap = near ? (u32)__mul24(near + 1, near + 1) : 0xffffffffu;
  const int rmax = max(i, n - 1 - i);
  const u32 lim = (u32)__mul24(rmax + 1, rmax + 1);
  const int hi_off = __mul24(n - 1, stride) + col;
  int oa = base, ob = base;
  u32 rr = 1u, inc = 3u;
  // two rows out per trip: four LDS reads in flight behind one wait and one exit test.  The second step may be
  // one more than the exit test would have allowed -- every candidate f(i +- r) + r^2 is a true upper bound, so
  // extra ones cannot change the minimum.  r^2 advances by additions (32-bit multiplies are quarter rate).
  while (rr < min(min(mx, lim), cap)) {
    const int oa1 = max(oa - stride, col), ob1 = min(ob + stride, hi_off);
    oa = max(oa1 - stride, col);
    ob = min(ob1 + stride, hi_off);
    const uint4 va = lds4(tile, oa1), vb = lds4(tile, ob1), vc = lds4(tile, oa), vd = lds4(tile, ob);
    const u32 rr2 = rr + inc;
    // a clamped row repeats a candidate already seen with a smaller r: harmless; INF32 + r^2 stays above every
    // finite value and below 2^31
    b0 = min(b0, min(min(va.x, vb.x) + rr, min(vc.x, vd.x) + rr2));
    b1 = min(b1, min(min(va.y, vb.y) + rr, min(vc.y, vd.y) + rr2));
    b2 = min(b2, min(min(va.z, vb.z) + rr, min(vc.z, vd.z) + rr2));
    b3 = min(b3, min(min(va.w, vb.w) + rr, min(vc.w, vd.w) + rr2));
    mx = max4(b0, b1, b2, b3);
    rr = rr2 + inc + 2u;
    inc += 4u;
  }
  open = rr < min(mx, lim);
  return make_uint4(b0, b1, b2, b3);
}

// the line minimum first: in a flat field nothing can beat the output's own f
__device__ __forceinline__ bool line_can_improve(const unsigned char* tile, const unsigned char* cm, int stride, int i,
                                                 int col, uint4& v0) {
  v0 = lds4(tile, __mul24(i, stride) + col);
  const uint4 c = lds4(cm, col);
  return (c.x + 1u < v0.x) | (c.y + 1u < v0.y) | (c.z + 1u < v0.z) | (c.w + 1u < v0.w);
}

// far phase: bb = the running minima of an output the near phase left open (all rows within `near` seen); returns the exact minima
__device__ __forceinline__ uint4 scan_far4(const unsigned char* tile, const unsigned char* bm, const unsigned char* cm,
                                           int stride, int n, int i, int col, uint4 bb) {
  (void)cm;
  u32 b0 = bb.x, b1 = bb.y, b2 = bb.z, b3 = bb.w;
  u32 mx = max4(b0, b1, b2, b3);
  const int hi_off = __mul24(n - 1, stride) + col;
  // block walk: blocks kb -+ j, nearest rows dl / du away (seeing the rows of the near phase again is harmless)
  const int kb = i >> 3, off = i & 7, nblk = (n + 7) >> 3;
  {  // the output's own block first (the near phase may have stopped short of its ends)
    int o = __mul24(8 * kb, stride) + col;
#pragma unroll 4
    for (int t = 0; t < 8; ++t) {
      const uint4 q = lds4(tile, min(o, hi_off));
      const u32 r2 = (u32)__mul24(t - off, t - off);
      b0 = min(b0, q.x + r2), b1 = min(b1, q.y + r2), b2 = min(b2, q.z + r2), b3 = min(b3, q.w + r2);
      o += stride;
    }
    mx = max4(b0, b1, b2, b3);
  }
  for (int j = 1;; ++j) {
    const int dl = off + 8 * j - 7, du = 8 * j - off;
    const u32 dl2 = (u32)__mul24(dl, dl), du2 = (u32)__mul24(du, du);
    const bool lo = kb - j >= 0 && dl2 < mx;
    const bool up = kb + j < nblk && du2 < mx;
    if (!(lo | up)) break;
    if (lo) {
      const uint4 m = lds4(bm, __mul24(kb - j, stride) + col);
      if ((m.x + dl2 < b0) | (m.y + dl2 < b1) | (m.z + dl2 < b2) | (m.w + dl2 < b3)) {
        int o = __mul24(8 * (kb - j) + 7, stride) + col;
        u32 r2 = dl2, ic = 2u * (u32)dl + 1u;
#pragma unroll 4
        for (int t = 0; t < 8; ++t) {
          const uint4 q = lds4(tile, o);
          b0 = min(b0, q.x + r2), b1 = min(b1, q.y + r2), b2 = min(b2, q.z + r2), b3 = min(b3, q.w + r2);
          o -= stride;
          r2 += ic;
          ic += 2u;
        }
      }
    }
    if (up) {
      const uint4 m = lds4(bm, __mul24(kb + j, stride) + col);
      if ((m.x + du2 < b0) | (m.y + du2 < b1) | (m.z + du2 < b2) | (m.w + du2 < b3)) {
        int o = __mul24(8 * (kb + j), stride) + col;
        u32 r2 = du2, ic = 2u * (u32)du + 1u;
#pragma unroll 4
        for (int t = 0; t < 8; ++t) {
          const uint4 q = lds4(tile, min(o, hi_off));  // past the last row: that row again, with a larger r
          b0 = min(b0, q.x + r2), b1 = min(b1, q.y + r2), b2 = min(b2, q.z + r2), b3 = min(b3, q.w + r2);
          o += stride;
          r2 += ic;
          ic += 2u;
        }
      }
    }
    mx = max4(b0, b1, b2, b3);
  }
  return make_uint4(b0, b1, b2, b3);
}

// exact min_p f(p) + (i-p)^2 for the 4 columns at byte offset `col` of row i
template <bool FAR>
__device__ __forceinline__ uint4 scan_line4(const unsigned char* tile, const unsigned char* bm, const unsigned char* cm,
                                            int stride, int n, int i, int col, int near) {
  bool open = false;
  uint4 bb;
  if (!FAR) return scan_near4(tile, stride, n, i, col, 0, open);
  if (line_can_improve(tile, cm, stride, i, col, bb)) bb = scan_near4(tile, stride, n, i, col, near, open);
  if (open) bb = scan_far4(tile, bm, cm, stride, n, i, col, bb);
  return bb;
}

// dz^2 of one tile row from the chunk's source bits and the nearest sources below / above the chunk
__device__ __forceinline__ void zy_fill_row(u32* row, int ZC, u64 bits, int below, int above, int zs, int ze,
                                            u64 inbox_mask, bool fast) {
  // two sweeps over the chunk (distance to the nearest source below / above), 4 voxels per LDS
  // access.  Columns outside the box get garbage that the y pass never stores nor mixes in (a
  // column only reads itself in other rows).
  const u32 BIGD = 1u << 20;  // "no source yet": stays >= BIGD after any number of +1 steps
  if (fast) {
    // rows that need no sweep: every in-box voxel of the chunk a source (the inside of unknown space: half of a
    // half-explored map), or no source anywhere on the row's z-line
    const bool all_src = bits == inbox_mask && inbox_mask != 0ull;
    const bool no_src = bits == 0ull && below < 0 && above < 0;
    if (all_src | no_src) {
      const u32 v = all_src ? 0u : INF32;
      const uint4 o = make_uint4(v, v, v, v);
      for (int zi = 0; zi < ZC; zi += 4) *reinterpret_cast<uint4*>(row + zi) = o;
      return;
    }
  }
  u32 d = below >= 0 ? (u32)(zs - 1 - below) : BIGD;
  for (int zi = 0; zi < ZC; zi += 4) {
    const u32 nib = (u32)(bits >> zi);
    uint4 o;
    d = (nib & 1u) ? 0u : d + 1u;
    o.x = d;
    d = (nib & 2u) ? 0u : d + 1u;
    o.y = d;
    d = (nib & 4u) ? 0u : d + 1u;
    o.z = d;
    d = (nib & 8u) ? 0u : d + 1u;
    o.w = d;
    *reinterpret_cast<uint4*>(row + zi) = o;
  }
  d = above >= 0 ? (u32)(above - (ze + 1)) : BIGD;
  for (int zi = ZC - 4; zi >= 0; zi -= 4) {
    const u32 nib = (u32)(bits >> zi);
    uint4 o = *reinterpret_cast<const uint4*>(row + zi);
    d = (nib & 8u) ? 0u : d + 1u;
    o.w = min(o.w, d);
    d = (nib & 4u) ? 0u : d + 1u;
    o.z = min(o.z, d);
    d = (nib & 2u) ? 0u : d + 1u;
    o.y = min(o.y, d);
    d = (nib & 1u) ? 0u : d + 1u;
    o.x = min(o.x, d);
    o.x = o.x >= BIGD ? INF32 : o.x * o.x;
    o.y = o.y >= BIGD ? INF32 : o.y * o.y;
    o.z = o.z >= BIGD ? INF32 : o.z * o.z;
    o.w = o.w >= BIGD ? INF32 : o.w * o.w;
    *reinterpret_cast<uint4*>(row + zi) = o;
  }
}

#define ZY_ROUNDS 2  // rows per lane whose plane words are fetched up front (y lines of up to 1024 voxels)
// bits of word k of a multi-word line that lie in positions [A, B]
__device__ __forceinline__ u64 range_mask(int k, int A, int B) {
  const int lo = max(A - 64 * k, 0), hi = min(B - 64 * k, 63);
  return lo <= hi ? bit_range(lo, hi - lo + 1) : 0ull;
}

template <int MODE, bool FAR>
__global__ void __launch_bounds__(512)
k_esdf_zy4(Geo g, Box3 b, const u64* __restrict__ infl, const u64* __restrict__ unk, u32* __restrict__ tmp,
           int ZC, int nzc, int z0a, int near, u32* __restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // [ylen][ZC] dz^2 or INF32 (u32: the scan loop adds without sentinel tests), ZC % 4 == 0, ZC <= 64; FAR: then
  // the block minima [ceil(ylen/8)][ZC] and the line minima [ZC]
  u32* tile = reinterpret_cast<u32*>(smem_raw);
  // XCD-aware order: workgroup i runs on XCD i % 8, and each XCD has its own L2.  The nzc chunk-blocks of
  // one x-slab read the same bit-plane lines and write interleaved pieces of the same tmp lines, so they
  // are given to the SAME XCD (slab x -> XCD x % 8) instead of being dealt round-robin over all eight.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int xrel = xcd + 8 * (slot / nzc);
  if (xrel > b.hi[0] - b.lo[0]) return;  // grid is padded to a multiple of 8 slabs
  const int x = b.lo[0] + xrel;
  const int zc0 = z0a + (slot % nzc) * ZC;
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int T = blockDim.x;
  // z pass: one lane per row of the chunk; chunk source bits live in one register
  const int zs = max(zc0, b.lo[2]), ze = min(zc0 + ZC - 1, b.hi[2]);  // in-box part of the chunk
  unsigned char* bm = smem_raw + (size_t)ylen * ZC * 4;
  u32* cm = reinterpret_cast<u32*>(bm + (size_t)((ylen + 7) >> 3) * ZC * 4);
  if (FAR && (int)threadIdx.x < ZC) cm[threadIdx.x] = INF32;
  const u64 inbox_mask = zs <= ze ? bit_range(zs - zc0, ze - zs + 1) : 0ull;
  const bool fastrow = (near >> 8) & 1;  // (FUELMI_ZY_FASTROW=0 switches the sweep-free rows off)
  near &= 0xff;
  // Source bits of the rows.  When the box's part of a z-line fits three plane words (z extents up to 129
  // voxels) every row takes ONE round of independent loads -- issued for all of the lane's rows before the first
  // is used -- instead of a chain of dependent ones (word pair of the chunk, then the searches below and above it):
  // the kernel is latency-bound and that chain was half of a workgroup's life.
  if (b.hi[2] - z0a + 1 <= 129 && ylen <= ZY_ROUNDS * T) {
    u64 L[ZY_ROUNDS][3];
#pragma unroll
    for (int r = 0; r < ZY_ROUNDS; ++r) {
      const int yi = threadIdx.x + r * T;
      L[r][0] = L[r][1] = L[r][2] = 0ull;
      if (yi < ylen && zs <= ze) {
        const long linebit = (long)x * g.nyz + (long)(b.lo[1] + yi) * g.nz;
        const long w0 = (linebit + z0a) >> 6, w1 = (linebit + b.hi[2]) >> 6;
        L[r][0] = src_word<MODE>(infl, unk, w0);
        if (w0 + 1 <= w1) L[r][1] = src_word<MODE>(infl, unk, w0 + 1);
        if (w0 + 2 <= w1) L[r][2] = src_word<MODE>(infl, unk, w0 + 2);
      }
    }
#pragma unroll
    for (int r = 0; r < ZY_ROUNDS; ++r) {
      const int yi = threadIdx.x + r * T;
      if (yi >= ylen) break;
      const long linebit = (long)x * g.nyz + (long)(b.lo[1] + yi) * g.nz;
      u64 bits = 0ull;
      int below = -1, above = -1;
      if (zs <= ze) {
        const u64 L0 = L[r][0], L1 = L[r][1], L2 = L[r][2];
        const int p0 = (int)((linebit + z0a) & 63) - z0a;  // position of voxel z in the 192-bit line = p0 + z
        const int pc = p0 + zc0, q = pc >> 6, sh = pc & 63;
        const u64 a = q == 0 ? L0 : (q == 1 ? L1 : L2), c = q == 0 ? L1 : (q == 1 ? L2 : 0ull);
        bits = sh ? ((a >> sh) | (c << (64 - sh))) : a;
        bits &= bit_range(zs - zc0, ze - zs + 1);
        if (zs - 1 >= b.lo[2]) {  // highest source below the chunk
          const int A = p0 + b.lo[2], B = p0 + zs - 1;
          const u64 m2 = L2 & range_mask(2, A, B), m1 = L1 & range_mask(1, A, B), m0 = L0 & range_mask(0, A, B);
          const int hp = m2 ? 191 - __builtin_clzll(m2) : (m1 ? 127 - __builtin_clzll(m1) : (m0 ? 63 - __builtin_clzll(m0) : -1));
          if (hp >= 0) below = hp - p0;
        }
        if (ze + 1 <= b.hi[2]) {  // lowest source above it
          const int A = p0 + ze + 1, B = p0 + b.hi[2];
          const u64 m0 = L0 & range_mask(0, A, B), m1 = L1 & range_mask(1, A, B), m2 = L2 & range_mask(2, A, B);
          const int lp = m0 ? __builtin_ctzll(m0) : (m1 ? 64 + __builtin_ctzll(m1) : (m2 ? 128 + __builtin_ctzll(m2) : -1));
          if (lp >= 0) above = lp - p0;
        }
      }
      zy_fill_row(tile + yi * ZC, ZC, bits, below, above, zs, ze, inbox_mask, fastrow);
    }
  } else {
    for (int yi = threadIdx.x; yi < ylen; yi += T) {
      const long linebit = (long)x * g.nyz + (long)(b.lo[1] + yi) * g.nz;
      u64 bits = 0ull;
      int below = -1, above = -1;
      if (zs <= ze) {
        u64 lo = src_word<MODE>(infl, unk, (linebit + zc0) >> 6), hi = src_word<MODE>(infl, unk, ((linebit + zc0) >> 6) + 1);
        int sh = (int)((linebit + zc0) & 63);
        bits = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        bits &= bit_range(zs - zc0, ze - zs + 1);
        below = line_src_down<MODE>(infl, unk, linebit, b.lo[2], zs - 1);
        above = line_src_up<MODE>(infl, unk, linebit, ze + 1, b.hi[2]);
      }
      zy_fill_row(tile + yi * ZC, ZC, bits, below, above, zs, ze, inbox_mask, fastrow);
    }
  }
  __syncthreads();
  if (FAR) {
    build_block_minima(smem_raw, bm, cm, ZC * 4, ZC >> 2, ylen);
    __syncthreads();
  }
  // y pass: one lane per 4 z-adjacent outputs
  const int G = ZC >> 2;
  const int total = ylen * G;
  const int stride = ZC * 4;
  const int dyi = T / G, dgi = T - dyi * G;
  // every 16th slab reports how many of its outputs are further than ESDF_FAR_D voxels from every source: the
  // host picks the FAR or the plain kernels for the NEXT update from that (esdf_update)
  const bool sampled = stat != nullptr && ((x & 15) == 0 || xrel == 0);  // (absolute slabs: the statistic is kept per place)
  int n_far = 0;
  const bool z_aligned = (b.lo[2] & 3) == 0 && (b.hi[2] & 3) == 3;
  int yi = threadIdx.x / G, gi = threadIdx.x - yi * G;
  for (int o = threadIdx.x; o < total; o += T) {
    const uint4 bb = scan_line4<FAR>(smem_raw, bm, reinterpret_cast<const unsigned char*>(cm), stride, ylen, yi, 16 * gi, near);
    if (sampled) n_far += __popcll(__ballot(min(min(bb.x, bb.y), min(bb.z, bb.w)) > ESDF_FAR_D * ESDF_FAR_D));
    const int z = zc0 + 4 * gi;
    u32* dst = tmp + (long)x * g.nyz + (long)(b.lo[1] + yi) * g.nz + z;
    // (z_aligned is uniform: the common 4-aligned box keeps ONE 16-byte store; folded into the per-lane test the
    // compiler merged the two paths into a 12-byte and a 4-byte store)
    if (z_aligned) {
      if (z >= b.lo[2] && z <= b.hi[2]) store16(dst, bb);  // (chunks may reach past the box)
    } else if (z >= b.lo[2] && z + 3 <= b.hi[2]) {
      *reinterpret_cast<uint4*>(dst) = bb;
    } else {
      if (z >= b.lo[2] && z <= b.hi[2]) dst[0] = bb.x;
      if (z + 1 >= b.lo[2] && z + 1 <= b.hi[2]) dst[1] = bb.y;
      if (z + 2 >= b.lo[2] && z + 2 <= b.hi[2]) dst[2] = bb.z;
      if (z + 3 >= b.lo[2] && z + 3 <= b.hi[2]) dst[3] = bb.w;
    }
    yi += dyi;
    gi += dgi;
    if (gi >= G) {
      gi -= G;
      ++yi;
    }
  }
  if (sampled && (threadIdx.x & 63) == 0) {
    u32* sg = stat + 2 * ((x >> 4) & (ESDF_NG - 1));  // group of 16 slabs this one stands for
    atomicAdd(sg, (u32)n_far);
    if (threadIdx.x == 0) atomicAdd(sg + 1, (u32)total);
  }
}

// ------------------------------------------------------------------------------------------------
// Packed plain family (round 4): the z/y pass on 16-bit lanes, two y-rows per lane.
// The u32 kernel above is issue-bound (~36 VALU instructions per voxel), so this one halves the
// instruction stream instead of the bytes: the tile holds dz^2 as u16, rows 2p and 2p+1 interleaved in one u32
// ("pair row" p: low half = row 2p, high half = row 2p+1), and every min / add of the scan is a v_pk_*_u16 that
// serves both rows.  A lane owns 8 outputs (rows 2p, 2p+1 x 4 z); one trip loads pair rows p-j and p+j (two
// ds_read_b128) and examines rows 2p-2j .. 2p+2j+1 for both outputs: straight halves are 2j away from both, swapped
// halves (op_sel, free) 2j-1 / 2j+1 away.  Saturating adds keep everything below 2^16; PK_INF = 255^2 marks
// "no source on this z-line" (a real dz is at most 254: the family is used for boxes of up to 255 voxels in z).
// A result below PK_INF is exact: every candidate f(p) + r^2 < 65025 was formed without saturation and every
// saturated one is >= 65025.  The rare outputs at or above it (further than ~255 voxels from every source, or no
// source in the slab) are recomputed in 32 bits from the same tile (pk_slow_col).
// ------------------------------------------------------------------------------------------------
#define PK_INF 65025u
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2_t pk_v(u32 v) { return __builtin_bit_cast(u16x2_t, v); }
__device__ __forceinline__ u32 pk_u(u16x2_t v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) { return pk_u(__builtin_elementwise_min(pk_v(a), pk_v(b))); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return pk_u(__builtin_elementwise_max(pk_v(a), pk_v(b))); }
__device__ __forceinline__ u32 pk_adds(u32 a, u32 b) { return pk_u(__builtin_elementwise_add_sat(pk_v(a), pk_v(b))); }
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return pk_u(pk_v(a) + pk_v(b)); }
__device__ __forceinline__ u32 pk_mul(u32 a, u32 b) { return pk_u(pk_v(a) * pk_v(b)); }
__device__ __forceinline__ u32 pk_swap(u32 a) {
  const u16x2_t v = pk_v(a);
  const u16x2_t r = {v.y, v.x};
  return pk_u(r);
}
__device__ __forceinline__ u32 pk_hmax(u32 a) { return max(a & 0xffffu, a >> 16); }
__device__ __forceinline__ u32 pk_hmin(u32 a) { return min(a & 0xffffu, a >> 16); }
__device__ __forceinline__ u32 pk_both(u32 v) { return v | (v << 16); }
__device__ __forceinline__ u32 pk_sat16(u32 v) { return min(v, 0xffffu); }

// chunk source bits and nearest sources of the box below / above the chunk (absolute z, -1: none) for one y-row
struct RowSrc {
  u64 bits;
  int below, above;
};
// The source bits of one z-line, ALIGNED: bit k = voxel z0a + k, bits outside the box's z range cleared.  Built once
// per line from NW + 1 plane words (a per-lane funnel shift); everything a chunk needs afterwards -- its own bits, the
// nearest source below / above it -- is windows and masks at UNIFORM positions (scalar masks, 32-bit vector work).
// The round-3 form (range masks over three unaligned words, per lane, in 64-bit arithmetic) cost ~250 VALU
// instructions per row and chunk: more than the two sweeps it feeds (SQ_INSTS_VALU, profiles/r04_zy_*).
template <int NW>
struct LineBits {
  u64 w[NW];
};
template <int MODE, int NW>
__device__ __forceinline__ LineBits<NW> line_load(const u64* __restrict__ infl, const u64* __restrict__ unk, long linebit,
                                                  int z0a, int zlo, int zhi) {
  const long bit0 = linebit + z0a;
  const long w0 = bit0 >> 6;
  const int sh = (int)(bit0 & 63);
  u64 L[NW + 1];
#pragma unroll
  for (int k = 0; k <= NW; ++k) L[k] = src_word<MODE>(infl, unk, w0 + k);  // (planes carry zeroed / masked-off margins)
  LineBits<NW> r;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int lo = max(zlo - z0a - 64 * k, 0), hi = min(zhi - z0a - 64 * k, 63);  // in-box bits of this word (uniform)
    const u64 mask = lo <= hi ? bit_range(lo, hi - lo + 1) : 0ull;
    r.w[k] = (sh ? ((L[k] >> sh) | (L[k + 1] << (64 - sh))) : L[k]) & mask;
  }
  return r;
}
// 32 bits of the line from bit s (uniform, multiple of 4)
template <int NW>
__device__ __forceinline__ u32 line_bits32(const LineBits<NW>& L, int s) {
  const int k = s >> 6, sh = s & 63;
  u64 lo = 0ull, hi = 0ull;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    if (i == k) lo = L.w[i];
    if (i == k + 1) hi = L.w[i];
  }
  return (u32)(sh ? ((lo >> sh) | (hi << (64 - sh))) : lo);
}
// lowest set bit at or above bit s / highest set bit below bit s (uniform s), or -1
template <int NW>
__device__ __forceinline__ int line_first_from(const LineBits<NW>& L, int s) {
  int pos = -1;
#pragma unroll
  for (int i = NW - 1; i >= 0; --i) {
    u64 m = L.w[i];
    if (64 * i + 63 < s)
      m = 0ull;
    else if (64 * i < s)
      m &= ~0ull << (s - 64 * i);
    if (m) pos = 64 * i + __builtin_ctzll(m);
  }
  return pos;
}
template <int NW>
__device__ __forceinline__ int line_last_below(const LineBits<NW>& L, int s) {
  int pos = -1;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    u64 m = L.w[i];
    if (64 * i >= s)
      m = 0ull;
    else if (64 * i + 64 > s)
      m &= (1ull << (s - 64 * i)) - 1ull;
    if (m) pos = 64 * i + 63 - __builtin_clzll(m);
  }
  return pos;
}
// chunk [zc0, zc0 + ZC) of an aligned line: its bits, the nearest sources of the box below and above it (absolute z)
template <int NW, bool BELOW>
__device__ __forceinline__ RowSrc row_src_line(const LineBits<NW>& L, int z0a, int zc0, int ZC) {
  RowSrc r;
  const int s = zc0 - z0a;
  r.bits = (u64)(line_bits32<NW>(L, s) & (ZC == 32 ? 0xffffffffu : ((1u << ZC) - 1u)));
  const int up = line_first_from<NW>(L, s + ZC);
  r.above = up >= 0 ? z0a + up : -1;
  r.below = -1;
  if (BELOW) {
    const int dn = line_last_below<NW>(L, s);
    r.below = dn >= 0 ? z0a + dn : -1;
  }
  return r;
}

// sign-extended bit zi of a per-lane word: 0 or 0xFFFFFFFF in ONE instruction (the builtin is canonicalised into
// and + compare + select)
__device__ __forceinline__ u32 bit_fill(u32 v, int zi) {
  u32 r;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(v), "s"(zi));
  return r;
}
// d0: packed distance of voxel zs - 1 of both rows to their nearest source below (255: none); returns the packed
// distance of the chunk's last voxel (capped at 255) -- the next chunk's d0 when one lane walks up a z-line.
// Two sweeps like zy_fill_row, but on both rows at once: "d = source ? 0 : d + 1" is a packed add and an AND with the
// pair's not-a-source mask of that z (built once, used by both sweeps).  The forward distances wait in the tile row
// itself (LDS) for the backward sweep: 20 registers fewer, which is what lets eight workgroups share a CU.
template <int G>
__device__ __forceinline__ u32 zy_fill_pair(u32* prow, u32 bA, u32 bB, u32 d0, int aboveA, int aboveB, int ze, u32 inbox, bool fast) {
  constexpr int ZC = 4 * G;
  const u32 full = ZC == 32 ? 0xffffffffu : ((1u << (ZC & 31)) - 1u);
  if (fast) {
    // rows that need no sweep: every in-box voxel of the chunk a source (the inside of unknown space), or no source
    // anywhere on the z-line.  Taken by whole waves only: a mixed wave would pay for both paths
    const bool all_src = bA == inbox && bB == inbox && inbox != 0u;
    const bool no_src = (bA | bB) == 0u && d0 == 0x00ff00ffu && aboveA < 0 && aboveB < 0;
    if (__ballot(!(all_src | no_src)) == 0ull) {
      const u32 v = all_src ? 0u : pk_both(PK_INF);
      const uint4 o = make_uint4(v, v, v, v);
#pragma unroll
      for (int k = 0; k < G; ++k) *reinterpret_cast<uint4*>(prow + 4 * k) = o;
      if (inbox != full) {  // (columns outside the box must read 0: they only stretch the scans otherwise)
        const u32 ib = inbox | (threadIdx.x & 0u);  // a per-lane copy: the masks below stay out of the scalar registers
        for (int zi = 0; zi < ZC; ++zi) prow[zi] &= bit_fill(ib, zi);
      }
      return all_src ? 0u : 0x00ff00ffu;
    }
  }
  const u32 nA = ~bA, nB = ~bB;
  u32 nm[ZC];
  u32 d = d0;
#pragma unroll
  for (int k = 0; k < G; ++k) {
    u32 f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int zi = 4 * k + q;
      nm[zi] = (bit_fill(nA, zi) & 0xffffu) | (bit_fill(nB, zi) & 0xffff0000u);  // (one bit-field insert)
      d = pk_add(d, 0x00010001u) & nm[zi];  // (at most 255 + 32: no carry between the halves)
      f[q] = d;
    }
    *reinterpret_cast<uint4*>(prow + 4 * k) = make_uint4(f[0], f[1], f[2], f[3]);
  }
  const u32 d_end = pk_min(d, 0x00ff00ffu);
  d = (aboveA >= 0 ? (u32)min(aboveA - (ze + 1), 255) : 255u) | ((aboveB >= 0 ? (u32)min(aboveB - (ze + 1), 255) : 255u) << 16);
#pragma unroll
  for (int k = G - 1; k >= 0; --k) {
    const uint4 fw = *reinterpret_cast<const uint4*>(prow + 4 * k);
    const u32 f[4] = {fw.x, fw.y, fw.z, fw.w};
    u32 o[4];
#pragma unroll
    for (int q = 3; q >= 0; --q) {
      d = pk_add(d, 0x00010001u) & nm[4 * k + q];
      const u32 t = pk_min(pk_min(f[q], d), 0x00ff00ffu);
      o[q] = pk_mul(t, t);
    }
    *reinterpret_cast<uint4*>(prow + 4 * k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  if (inbox != full) {
    const u32 ib = inbox | (threadIdx.x & 0u);
    for (int zi = 0; zi < ZC; ++zi) prow[zi] &= bit_fill(ib, zi);
  }
  return d_end;
}
__device__ __forceinline__ u32 pk_below_d0(const RowSrc& A, const RowSrc& B, int zs) {
  return (A.below >= 0 ? (u32)min(zs - 1 - A.below, 255) : 255u) | ((B.below >= 0 ? (u32)min(zs - 1 - B.below, 255) : 255u) << 16);
}

// exact 32-bit value of one output the packed scan left at or above PK_INF (row of the y line, column zcol of the
// chunk): the same pruned scan over the 16-bit tile, rows [0, ylen)
__device__ __noinline__ u32 pk_slow_col(const u32* tile, int ZC, int ylen, int row, int zcol) {
  u32 best = INF32;
  const int rmax = max(row, ylen - 1 - row);
  for (int r = 0; r <= rmax && (u32)(r * r) < best; ++r) {
    const u32 rr = (u32)(r * r);
    if (row - r >= 0) {
      const int q = row - r;
      const u32 v = (tile[(q >> 1) * ZC + zcol] >> (16 * (q & 1))) & 0xffffu;
      if (v < PK_INF) best = min(best, v + rr);
    }
    if (r && row + r < ylen) {
      const int q = row + r;
      const u32 v = (tile[(q >> 1) * ZC + zcol] >> (16 * (q & 1))) & 0xffffu;
      if (v < PK_INF) best = min(best, v + rr);
    }
  }
  return best;
}

// the 8 outputs (rows 2p, 2p + 1 x 4 z of group gi) of one lane from the packed tile at `tile_b`
// TILE_SLOW: outputs at or above PK_INF are recomputed from the tile itself (z/y pass: the tile holds every finite value
// exactly); false: they are left to the caller (x pass: PK_INF in its tile may stand for a larger finite value)
template <int G, bool TILE_SLOW = true>
__device__ __forceinline__ void pk_scan8(const unsigned char* tile_b, int p, int gi, int npair, int ylen, bool any_src,
                                         bool sampled, int& n_far, uint4& ra, uint4& rb) {
  constexpr int ZC = 4 * G;
  constexpr int stride = ZC * 4;
  const u32* tile = reinterpret_cast<const u32*>(tile_b);
    const int col = 16 * gi, base = __mul24(p, stride) + col;
    if (!any_src) {  // nothing in this slab's chunk is a source: "no source in the box" everywhere
      ra = rb = make_uint4(INF32, INF32, INF32, INF32);
      if (sampled) n_far += __popcll(__ballot(1));
    } else {
      const uint4 v = lds4(tile_b, base);
      // the partner row is one row away
      u32 b0 = pk_min(v.x, pk_adds(pk_swap(v.x), 0x00010001u)), b1 = pk_min(v.y, pk_adds(pk_swap(v.y), 0x00010001u));
      u32 b2 = pk_min(v.z, pk_adds(pk_swap(v.z), 0x00010001u)), b3 = pk_min(v.w, pk_adds(pk_swap(v.w), 0x00010001u));
      u32 mx = pk_hmax(pk_max(pk_max(b0, b1), pk_max(b2, b3)));
      const int hi_off = __mul24(npair - 1, stride) + col;
      const int jmax = max(p, npair - 1 - p);
      int od = base, ou = base;
      // trip j: pair rows p - j, p + j = rows 2p - 2j .. 2p + 2j + 1; ro2 = (2j-1)^2 is the first radius not seen yet
      u32 ro2 = 1u, j8 = 8u;
      for (int j = 1; j <= jmax && ro2 < mx; ++j) {
        od = max(od - stride, col);
        ou = min(ou + stride, hi_off);
        const uint4 pd = lds4(tile_b, od), pu = lds4(tile_b, ou);
        const u32 rn2 = ro2 + j8;                  // (2j+1)^2
        const u32 re2 = (ro2 + rn2 - 2u) >> 1;     // (2j)^2
        const u32 ke = pk_both(pk_sat16(re2));
        const u32 k1 = pk_sat16(ro2) | (pk_sat16(rn2) << 16), k2 = pk_sat16(rn2) | (pk_sat16(ro2) << 16);
        // (a clamped pair row repeats candidates already seen with smaller radii: harmless)
        b0 = pk_min(b0, pk_min(pk_adds(pk_min(pd.x, pu.x), ke), pk_min(pk_adds(pk_swap(pd.x), k1), pk_adds(pk_swap(pu.x), k2))));
        b1 = pk_min(b1, pk_min(pk_adds(pk_min(pd.y, pu.y), ke), pk_min(pk_adds(pk_swap(pd.y), k1), pk_adds(pk_swap(pu.y), k2))));
        b2 = pk_min(b2, pk_min(pk_adds(pk_min(pd.z, pu.z), ke), pk_min(pk_adds(pk_swap(pd.z), k1), pk_adds(pk_swap(pu.z), k2))));
        b3 = pk_min(b3, pk_min(pk_adds(pk_min(pd.w, pu.w), ke), pk_min(pk_adds(pk_swap(pd.w), k1), pk_adds(pk_swap(pu.w), k2))));
        mx = pk_hmax(pk_max(pk_max(b0, b1), pk_max(b2, b3)));
        ro2 = rn2;
        j8 += 8u;
      }
      if (sampled) n_far += __popcll(__ballot(pk_hmin(pk_min(pk_min(b0, b1), pk_min(b2, b3))) > ESDF_FAR_D * ESDF_FAR_D));
      ra = make_uint4(b0 & 0xffffu, b1 & 0xffffu, b2 & 0xffffu, b3 & 0xffffu);
      rb = make_uint4(b0 >> 16, b1 >> 16, b2 >> 16, b3 >> 16);
      if (TILE_SLOW && mx >= PK_INF) {  // rare: an output out of the 16-bit range
        const int zc = 4 * gi, rA = 2 * p, rB = 2 * p + 1;
        if (ra.x >= PK_INF) ra.x = pk_slow_col(tile, ZC, ylen, rA, zc);
        if (ra.y >= PK_INF) ra.y = pk_slow_col(tile, ZC, ylen, rA, zc + 1);
        if (ra.z >= PK_INF) ra.z = pk_slow_col(tile, ZC, ylen, rA, zc + 2);
        if (ra.w >= PK_INF) ra.w = pk_slow_col(tile, ZC, ylen, rA, zc + 3);
        if (rB < ylen) {
          if (rb.x >= PK_INF) rb.x = pk_slow_col(tile, ZC, ylen, rB, zc);
          if (rb.y >= PK_INF) rb.y = pk_slow_col(tile, ZC, ylen, rB, zc + 1);
          if (rb.z >= PK_INF) rb.z = pk_slow_col(tile, ZC, ylen, rB, zc + 2);
          if (rb.w >= PK_INF) rb.w = pk_slow_col(tile, ZC, ylen, rB, zc + 3);
        }
      }
    }
}

// The y pass of this update has counted its far outputs per group of 16 x-slabs (stat[2 g], stat[2 g + 1]: far
// outputs, outputs of the sampled slab); the first workgroup of the x pass hands the sampled groups to the host through
// pinned memory and clears them.  No fence, no synchronisation (a system-scope release here makes this workgroup write
// the L2 back while the rest of the kernel fills it: the pass went from 35 to 90 us).  Every entry of the host table
// is ONE aligned 8-byte word -- far outputs (24 bits) | sampled outputs (24 bits) | the update's tag (16 bits, never
// 0) -- so an entry is self-consistent however the store races with the host's read, tables of several updates may
// land between two host looks, and the host never writes the table (the round-3 epoch + checksum over a
// host-cleared table could lock itself out of the far-field kernels for good).  The statistic picks a kernel family,
// it never changes a result.
__device__ __forceinline__ void forward_stat(u32* stat, volatile u32* h_stat_v) {
  if (stat == nullptr || blockIdx.x != 0) return;
  unsigned long long* h_stat = reinterpret_cast<unsigned long long*>(const_cast<u32*>(h_stat_v));  // (plain posted stores)
  const u32 e = stat[2 * ESDF_NG] + 1u;
  const unsigned long long tag = (unsigned long long)(e % 65535u + 1u) << 48;
  for (int g = threadIdx.x; g < ESDF_NG; g += blockDim.x) {
    uint2 p = *reinterpret_cast<const uint2*>(stat + 2 * g);  // (far outputs, outputs of the sampled slab)
    if (p.y) {  // only the groups this update sampled cross the bus; the host keeps the others
      *reinterpret_cast<uint2*>(stat + 2 * g) = make_uint2(0u, 0u);
      while (p.y >> 24) p.x >>= 1, p.y >>= 1;  // (the ratio is what matters)
      h_stat[g] = (unsigned long long)p.x | ((unsigned long long)p.y << 24) | tag;
    }
  }
  __syncthreads();  // (every lane has read the epoch)
  if (threadIdx.x == 0) stat[2 * ESDF_NG] = e;
}

// the 4 outputs of one lane: distance_buffer_ values (OUT 0) or the negative pass merged in place (OUT 1)
template <int OUT>
__device__ __forceinline__ void x_store4(float* dst, int z, int zlo, int zhi, uint4 bb, float resf) {
  if (OUT == 0) {
    if ((zlo & 3) == 0 && (zhi & 3) == 3) {  // uniform: one 16-byte store (see k_esdf_zy4); z is inside the box
      store16(dst, make_uint4(__float_as_uint(esdf_out(bb.x, resf)), __float_as_uint(esdf_out(bb.y, resf)),
                              __float_as_uint(esdf_out(bb.z, resf)), __float_as_uint(esdf_out(bb.w, resf))));
    } else if (z >= zlo && z + 3 <= zhi) {
      *reinterpret_cast<float4*>(dst) =
          make_float4(esdf_out(bb.x, resf), esdf_out(bb.y, resf), esdf_out(bb.z, resf), esdf_out(bb.w, resf));
    } else {
      if (z >= zlo && z <= zhi) dst[0] = esdf_out(bb.x, resf);
      if (z + 1 >= zlo && z + 1 <= zhi) dst[1] = esdf_out(bb.y, resf);
      if (z + 2 >= zlo && z + 2 <= zhi) dst[2] = esdf_out(bb.z, resf);
      if (z + 3 >= zlo && z + 3 <= zhi) dst[3] = esdf_out(bb.w, resf);
    }
  } else {
    if (z >= zlo && z <= zhi) dst[0] = esdf_merge_neg(dst[0], bb.x, resf);
    if (z + 1 >= zlo && z + 1 <= zhi) dst[1] = esdf_merge_neg(dst[1], bb.y, resf);
    if (z + 2 >= zlo && z + 2 <= zhi) dst[2] = esdf_merge_neg(dst[2], bb.z, resf);
    if (z + 3 >= zlo && z + 3 <= zhi) dst[3] = esdf_merge_neg(dst[3], bb.w, resf);
  }
}

// x pass, 4*SEGS columns (SEGS lanes x 4) per tile row; SEGS = 8: a wave covers 8 x-rows x 32 columns
// (128 B per row); SEGS = 4 halves the LDS tile for long x lines so that several workgroups still
// share a CU
template <int OUT, int SEGS, bool FAR>
__global__ void __launch_bounds__(1024)
k_esdf_x4(Geo g, Box3 b, const u32* __restrict__ tmp, float* __restrict__ dist, int z0a, int zlen_a, int near,
          u32* stat, volatile u32* h_stat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  forward_stat(stat, h_stat);
  uint4* tile = reinterpret_cast<uint4*>(smem_raw);  // [xlen][SEGS] of uint4 (FAR: + block / line minima)
  const int xlen = b.hi[0] - b.lo[0] + 1;
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int ncol = ylen * zlen_a;
  const int seg = threadIdx.x & (SEGS - 1);
  const int row0 = threadIdx.x / SEGS;  // 0..rows-1
  const int rows = blockDim.x / SEGS;
  const int col = blockIdx.x * (4 * SEGS) + seg * 4;
  const bool valid = col < ncol;
  const int yy = valid ? col / zlen_a : 0;
  const int z = z0a + (valid ? col - yy * zlen_a : 0);
  const long coloff = (long)(b.lo[1] + yy) * g.nz + z;
  const uint4 inf4 = make_uint4(INF32, INF32, INF32, INF32);
  const float resf = (float)g.res;
  unsigned char* bm = smem_raw + (size_t)xlen * SEGS * 16;
  u32* cm = reinterpret_cast<u32*>(bm + (size_t)((xlen + 7) >> 3) * SEGS * 16);
#pragma unroll 4
  for (int xi = row0; xi < xlen; xi += rows)
    tile[xi * SEGS + seg] = valid ? *reinterpret_cast<const uint4*>(tmp + (long)(b.lo[0] + xi) * g.nyz + coloff) : inf4;
  if (FAR && (int)threadIdx.x < 4 * SEGS) cm[threadIdx.x] = INF32;
  __syncthreads();
  if (FAR) {
    build_block_minima(smem_raw, bm, cm, SEGS * 16, SEGS, xlen);
    __syncthreads();
  }
  if (!valid) return;
  for (int xi = row0; xi < xlen; xi += rows) {
    const uint4 bb = scan_line4<FAR>(smem_raw, bm, reinterpret_cast<const unsigned char*>(cm), SEGS * 16, xlen, xi, 16 * seg, near);
    x_store4<OUT>(dist + (long)(b.lo[0] + xi) * g.nyz + coloff, z, b.lo[2], b.hi[2], bb, resf);
  }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device, not once per launch
static hipError_t lds_attr_once(const void* fn, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;  // (function, device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  for (const auto& d : done)
    if (d.first == fn && d.second == dev) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.push_back({fn, dev});
  return e;
}

// "this kernel variant does not fit this box" (LDS tables, line length): the caller takes the next variant.  Positive,
// so that it can never be taken for one of the negative FUELMI_E* codes
enum { ESDF_NO_FIT = 1 };

// the device and the pinned-host tables of the far-output statistic
template <int OUT>
static u32* esdf_stat_dev(fuelmi_map* m) {
  return OUT == 0 ? m->esdf_stat : nullptr;
}
static volatile u32* esdf_stat_host(fuelmi_map* m) { return m->h_esdf_stat; }

static inline bool use_vec4(const Geo& g, int xlen) { return (g.nz % 4) == 0 && (size_t)xlen * 64 <= 150 * 1024; }

template <int MODE, bool FAR>
static int launch_zy4(fuelmi_map* m, const Box3& b) {
  const Geo& g = m->g;
  const int xlen = b.hi[0] - b.lo[0] + 1, ylen = b.hi[1] - b.lo[1] + 1;
  const int z0a = b.lo[2] & ~3, z1a = b.hi[2] | 3;
  const int zlen_a = z1a - z0a + 1;
  // ~20 z per chunk amortises the per-row bit fetches (measured: 400 rows x 20 z = 32 KB on G400,
  // 800 rows x 20 z = 64 KB on G800 are the optima; bigger tiles lose occupancy, smaller ones repeat the
  // row prologue)
  const int budget = std::min(std::max(32 * 1024, ylen * 4 * 20), 80 * 1024);
  int zc_max = std::max(4, (budget / (4 * ylen)) & ~3);
  zc_max = std::min(zc_max, std::min(zlen_a, 64));
  int nzc = (zlen_a + zc_max - 1) / zc_max;
  int ZC = (((zlen_a + nzc - 1) / nzc) + 3) & ~3;
  nzc = (zlen_a + ZC - 1) / ZC;
  size_t lds = (size_t)(FAR ? ylen + ((ylen + 7) >> 3) + 1 : ylen) * ZC * sizeof(u32);  // tile (+ block and line minima)
  if (lds > 160 * 1024) {
    if (FAR) return ESDF_NO_FIT;  // (the far-field tables do not fit beside the tile: the caller takes the plain kernel)
    fuelmi_set_error("ESDF y-line of %d voxels does not fit the LDS tile", ylen);
    return FUELMI_ELIMIT;
  }
  if (lds > 64 * 1024)
    HIPCHK(lds_attr_once(reinterpret_cast<const void*>(&k_esdf_zy4<MODE, FAR>), 160 * 1024));
  STAGE_LAUNCH(m, (k_esdf_zy4<MODE, FAR>), ((xlen + 7) / 8) * 8 * nzc, 512, lds, g, b, (const u64*)m->infl_bits.p, (const u64*)m->unk_bits.p,
               m->esdf_tmp, ZC, nzc, z0a, esdf_near() | (zy_fastrow() ? 256 : 0),
               MODE == 2 ? nullptr : esdf_stat_dev<0>(m));
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

// FUELMI_ZY_TIMING: a pass has left four 100 MHz stamps per workgroup (start, tile filled, barrier passed, done)
static int pass_timing_report(fuelmi_map* m, unsigned long long* dbg, int grid, const char* what) {
  HIPCHK(hipStreamSynchronize(m->stream));
  std::vector<unsigned long long> d((size_t)grid * 8);
  HIPCHK(hipMemcpy(d.data(), dbg, d.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(hipFree(dbg));
  unsigned long long t0 = ~0ull, t1 = 0;
  double ph[3] = {0, 0, 0}, life = 0;
  int n = 0;
  for (int w = 0; w < grid; ++w) {
    const unsigned long long* r = &d[(size_t)w * 8];
    if (!r[0] || !r[3]) continue;
    t0 = std::min(t0, r[0]), t1 = std::max(t1, r[3]);
    ph[0] += (double)(r[1] - r[0]), ph[1] += (double)(r[2] - r[1]), ph[2] += (double)(r[3] - r[2]), life += (double)(r[3] - r[0]);
    ++n;
  }
  int alive = 0, early = 0;
  const unsigned long long mid = t0 + (t1 - t0) / 2;
  for (int w = 0; w < grid; ++w) {
    const unsigned long long* r = &d[(size_t)w * 8];
    if (!r[0] || !r[3]) continue;
    if (r[0] <= mid && r[3] >= mid) ++alive;
    if (r[0] - t0 < 100) ++early;
  }
  std::vector<double> st, en;
  for (int w = 0; w < grid; ++w) {
    const unsigned long long* r = &d[(size_t)w * 8];
    if (!r[0] || !r[3]) continue;
    st.push_back((double)(r[0] - t0) / 100.0), en.push_back((double)(r[3] - t0) / 100.0);
  }
  std::sort(st.begin(), st.end());
  std::sort(en.begin(), en.end());
  int peak = 0;
  {
    size_t i = 0, j = 0;
    int cur = 0;
    while (i < st.size()) {
      if (st[i] <= en[j]) ++cur, ++i, peak = std::max(peak, cur);
      else --cur, ++j;
    }
  }
  {  // the slowest tenth of the workgroups: where THEIR life goes, and which blocks they are
    std::vector<std::pair<double, int>> lv;
    for (int w = 0; w < grid; ++w) {
      const unsigned long long* r = &d[(size_t)w * 8];
      if (!r[0] || !r[3]) continue;
      lv.push_back({(double)(r[3] - r[0]) / 100.0, w});
    }
    std::sort(lv.begin(), lv.end());
    const size_t k0 = lv.size() - std::max<size_t>(1, lv.size() / 10);
    double f = 0, bb = 0, sc = 0, lf = 0;
    for (size_t k = k0; k < lv.size(); ++k) {
      const unsigned long long* r = &d[(size_t)lv[k].second * 8];
      f += (double)(r[1] - r[0]), bb += (double)(r[2] - r[1]), sc += (double)(r[3] - r[2]), lf += lv[k].first;
    }
    const double nn = (double)(lv.size() - k0) * 100.0;
    std::fprintf(stderr, "[slowest tenth: fill %.2f barrier %.2f scan %.2f life %.2f us; slowest blocks:", f / nn, bb / nn, sc / nn, lf * 100.0 / nn);
    for (size_t k = lv.size() - std::min<size_t>(8, lv.size()); k < lv.size(); ++k) std::fprintf(stderr, " %d(%.1f)", lv[k].second, lv[k].first);
    std::fprintf(stderr, "]\n");
  }
  auto pct = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(f * (double)v.size()))]; };
  std::fprintf(stderr, "[%s: %d workgroups, span %.2f us; per workgroup: fill %.2f barrier %.2f scan %.2f life %.2f us; "
               "alive at mid-span %d (peak %d), started in the first us %d; starts 25/50/75/100 %%: %.1f %.1f %.1f %.1f us, ends 25/50/75: %.1f %.1f %.1f\n",
               what, n, (double)(t1 - t0) / 100.0, ph[0] / n / 100.0,
               ph[1] / n / 100.0, ph[2] / n / 100.0, life / n / 100.0, alive, peak, early, pct(st, 0.25), pct(st, 0.5), pct(st, 0.75),
               pct(st, 1.0), pct(en, 0.25), pct(en, 0.5), pct(en, 0.75));
  return FUELMI_OK;
}

template <int OUT, int SEGS, bool FAR>
static int launch_x4s(fuelmi_map* m, const Box3& b) {
  const Geo& g = m->g;
  const int xlen = b.hi[0] - b.lo[0] + 1, ylen = b.hi[1] - b.lo[1] + 1;
  const int z0a = b.lo[2] & ~3, z1a = b.hi[2] | 3;
  const int zlen_a = z1a - z0a + 1;
  const size_t lds = (size_t)(FAR ? xlen + ((xlen + 7) >> 3) + 1 : xlen) * SEGS * 4 * sizeof(u32);
  if (lds > 160 * 1024) return ESDF_NO_FIT;  // (FAR only: the longest lines, > 2270 voxels, scan without the far phase)
  if (lds > 64 * 1024)
    HIPCHK(lds_attr_once(reinterpret_cast<const void*>(&k_esdf_x4<OUT, SEGS, FAR>), 160 * 1024));
  int ncol = ylen * zlen_a;
  const int threads = 1024;
  STAGE_LAUNCH(m, (k_esdf_x4<OUT, SEGS, FAR>), (ncol + 4 * SEGS - 1) / (4 * SEGS), threads, lds, g, b, (const u32*)m->esdf_tmp,
               m->dist, z0a, zlen_a, esdf_near(), esdf_stat_dev<OUT>(m), esdf_stat_host(m));
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}
// ------------------------------------------------------------------------------------------------
// Round 5: the hand-over between the two packed passes is 16-bit and tile-contiguous.
// Rounds 2-4 stored the y-pass result as u32 at the voxel's own address (64 MB on the 400^2 x 100 map) and the x pass
// fetched its 32-column tile as 400 pieces of 128 B at a 160 KB stride, saturating every value to 16 bits on the way
// in: 58-75 % of an x-pass workgroup's life was that fill (profiles/r04_reference_order_timing.txt).  Now the z/y pass
// writes exactly what the x pass keeps in LDS:
//   tmp16[tile t][x-pair q][seg 0..7] : uint4 = 4 z-adjacent columns, each u32 = (x-row 2q | x-row 2q+1 << 16),
//   columns enumerated as in the x pass (s = y * nseg + zseg, tile = s >> 3, seg = s & 7),
// so an x-pass workgroup copies ONE contiguous block of npx * 128 B (25.6 KB for 400 rows) into LDS and scans.  A u32
// of that layout holds two x-slabs, so a z/y workgroup owns a slab PAIR: two tiles in LDS (lanes of the lower half fill
// slab 2q, the upper half slab 2q+1), one fused scan loop per lane over both tiles (two independent LDS chains in
// flight), two 16-byte stores per lane and trip where the u32 form needed four.
// (Also tried this round: the z pass as a kernel of its own writing one byte per voxel, so that the fill of this
// kernel becomes a 32-byte load and 16 squares per row pair and chunk.  Parity-green; the z/y kernel 30 -> 24 us on the
// 400^2 x 100 map, the byte kernel 14 us: its floor is a 16 MB store, and the z/y pass is no shorter than 24 us
// with a free fill -- the tail of the kernel is the SCAN of the workgroups in the explored half.  Removed.)
// Values: v < PK_INF exact; PK_INF = "finite, at least 65025" -- the exact u32 then sits in the WIDE plane (esdf_tmp, the
// voxel's own address; written only for such outputs: further than 255 voxels from every source of their slab);
// 0xFFFF = no source in the slab.  Columns of the aligned z range outside the box hold 0 (they only must not stretch a
// lane's shared scan loop).  Whether ANY slab of the box holds a source crosses to the x pass as one word stamped with
// the update's serial number (a slab pair with a source writes it; nobody clears it).
// ------------------------------------------------------------------------------------------------
#define PK_NOSRC 0xFFFFu
extern "C" __device__ int __ockl_wgred_or_i32(int a);  // workgroup-wide OR of a full int (what __syncthreads_or wraps)

// fused scan of item (p, gi) over the two tiles of a slab pair: the packed minima of rows 2p / 2p + 1 x 4 z of both
// slabs; want = bit 0 / 1: slab A / B holds a source at all (the other's tile is all PK_INF and must not keep the
// loop running)
template <int G>
__device__ __forceinline__ void pk_scan16(const unsigned char* tA, const unsigned char* tB, int p, int gi, int npair, int want,
                                          u32 (&a)[4], u32 (&c)[4]) {
  constexpr int stride = 16 * G;
  const int col = 16 * gi, base = __mul24(p, stride) + col;
  const uint4 vA = lds4(tA, base), vB = lds4(tB, base);
  const u32 one = 0x00010001u;
  a[0] = pk_min(vA.x, pk_adds(pk_swap(vA.x), one)), a[1] = pk_min(vA.y, pk_adds(pk_swap(vA.y), one));
  a[2] = pk_min(vA.z, pk_adds(pk_swap(vA.z), one)), a[3] = pk_min(vA.w, pk_adds(pk_swap(vA.w), one));
  c[0] = pk_min(vB.x, pk_adds(pk_swap(vB.x), one)), c[1] = pk_min(vB.y, pk_adds(pk_swap(vB.y), one));
  c[2] = pk_min(vB.z, pk_adds(pk_swap(vB.z), one)), c[3] = pk_min(vB.w, pk_adds(pk_swap(vB.w), one));
  const u32 mA = (want & 1) ? 0xffffffffu : 0u, mB = (want & 2) ? 0xffffffffu : 0u;
  u32 mx = pk_hmax(pk_max(pk_max(pk_max(a[0], a[1]), pk_max(a[2], a[3])) & mA, pk_max(pk_max(c[0], c[1]), pk_max(c[2], c[3])) & mB));
  const int hi_off = __mul24(npair - 1, stride) + col;
  const int jmax = max(p, npair - 1 - p);
  int od = base, ou = base;
  u32 ro2 = 1u, j8 = 8u;
  for (int j = 1; j <= jmax && ro2 < mx; ++j) {
    od = max(od - stride, col);
    ou = min(ou + stride, hi_off);
    const uint4 dA = lds4(tA, od), uA = lds4(tA, ou), dB = lds4(tB, od), uB = lds4(tB, ou);
    const u32 rn2 = ro2 + j8;               // (2j+1)^2
    const u32 re2 = (ro2 + rn2 - 2u) >> 1;  // (2j)^2
    const u32 ke = pk_both(pk_sat16(re2));
    const u32 k1 = pk_sat16(ro2) | (pk_sat16(rn2) << 16), k2 = pk_sat16(rn2) | (pk_sat16(ro2) << 16);
#define PK16_STEP(B, D, U) B = pk_min(B, pk_min(pk_adds(pk_min(D, U), ke), pk_min(pk_adds(pk_swap(D), k1), pk_adds(pk_swap(U), k2))))
    PK16_STEP(a[0], dA.x, uA.x);
    PK16_STEP(a[1], dA.y, uA.y);
    PK16_STEP(a[2], dA.z, uA.z);
    PK16_STEP(a[3], dA.w, uA.w);
    PK16_STEP(c[0], dB.x, uB.x);
    PK16_STEP(c[1], dB.y, uB.y);
    PK16_STEP(c[2], dB.z, uB.z);
    PK16_STEP(c[3], dB.w, uB.w);
#undef PK16_STEP
    mx = pk_hmax(pk_max(pk_max(pk_max(a[0], a[1]), pk_max(a[2], a[3])) & mA, pk_max(pk_max(c[0], c[1]), pk_max(c[2], c[3])) & mB));
    ro2 = rn2;
    j8 += 8u;
  }
}

// uint4 index of (tile, x-pair q, segment) in tmp16: a tile's rows are contiguous ([tile][q][8]: the x pass copies one
// block per tile).  (Rows of 8 or 64 x-pairs adjacent instead -- better locality for the z/y pass's stores -- left the z/y
// pass where it was and cost the x pass a third on the 800^2 x 200 map: its fill wants ONE run per tile.)
__device__ __forceinline__ size_t pk2_index(int tile, int q, int seg, int npx) { return ((size_t)tile * npx + q) * 8 + seg; }

// 16-bit hand-over value of one exact y-pass result; the exact value goes to the wide plane when it does not fit
__device__ __forceinline__ u32 pk2_encode(u32 exact, u32* wide_at, bool in_box) {
  if (exact >= INF32) return PK_NOSRC;
  if (exact < PK_INF) return exact;
  if (in_box) *wide_at = exact;
  return PK_INF;
}

// The z extent is cut into chunks of g = 8, 4, 2 or 1 z-segments (a segment = 4 voxels) that add up to it exactly
// (100 voxels: 8 + 8 + 8 + 1 segments); a column tile of the x pass is 8 / g y-rows x the g segments of ONE chunk, so
// that every 128-byte row of tmp16 -- (tile, x-pair) -- is written whole, by adjacent lanes of one wave of the chunk's
// workgroup.  (First version of the round: tiles of 8 consecutive segments of the (y, z) enumeration, chunks of 5
// segments.  Every tile row was then assembled from 80-byte pieces of two or three workgroups, and a wave's store
// touched 13 lines 640 KB apart: the z/y pass of the 800^2 x 200 map went from 169 to 297 us.)
// (struct Pk2Chunks: fuelmi_internal.h)

template <int MODE, int G, int NW>
__device__ __forceinline__ void zy_pk2_body(const Geo& g, const Box3& b, const u64* __restrict__ infl, const u64* __restrict__ unk,
                                            uint4* __restrict__ tmp16, u32* __restrict__ wide, int z0a, int seg0, int tile0, int tstride, int gsh,
                                            int off, int npx, int q, int fastrow, u32* __restrict__ stat, u32* __restrict__ src_flag, u32 serial,
                                            unsigned long long* stamp, unsigned char* smem_raw) {
  constexpr int ZC = 4 * G;
  const int xA = b.lo[0] + 2 * q, xB = min(xA + 1, b.hi[0]);  // (odd line: the last pair repeats its slab, like the x pass's tile)
  const int zc0 = z0a + 4 * seg0;
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int npair = (ylen + 1) >> 1;
  const int T = blockDim.x, TH = T >> 1;
  const int half = (int)threadIdx.x >= TH ? 1 : 0;
  const int x = half ? xB : xA;
  unsigned char* tA = smem_raw;                            // [npair][ZC] u32: low half y-row 2p, high half 2p + 1
  unsigned char* tB = smem_raw + (size_t)npair * ZC * 4;
  u32* tile = reinterpret_cast<u32*>(half ? tB : tA);
  const int zs = max(zc0, b.lo[2]), ze = min(zc0 + ZC - 1, b.hi[2]);
  const u32 inbox = zs <= ze ? (u32)bit_range(zs - zc0, ze - zs + 1) : 0u;
  bool has_src = false;
  for (int p = (int)threadIdx.x - half * TH; p < npair; p += TH) {
    const int yA = 2 * p, yB = min(2 * p + 1, ylen - 1);
    const long lbA = (long)x * g.nyz + (long)(b.lo[1] + yA) * g.nz;
    const long lbB = (long)x * g.nyz + (long)(b.lo[1] + yB) * g.nz;
    RowSrc A, B;
    A.bits = B.bits = 0ull;
    A.below = A.above = B.below = B.above = -1;
    if (zs <= ze) {
      const LineBits<NW> LA = line_load<MODE, NW>(infl, unk, lbA, z0a, b.lo[2], b.hi[2]);
      const LineBits<NW> LB = line_load<MODE, NW>(infl, unk, lbB, z0a, b.lo[2], b.hi[2]);
      A = row_src_line<NW, true>(LA, z0a, zc0, ZC);
      B = row_src_line<NW, true>(LB, z0a, zc0, ZC);
    }
    has_src |= (A.bits | B.bits) != 0ull || A.below >= 0 || A.above >= 0 || B.below >= 0 || B.above >= 0;
    (void)zy_fill_pair<G>(tile + p * ZC, (u32)A.bits, (u32)B.bits, pk_below_d0(A, B, zs), A.above, B.above, ze, inbox, (fastrow & 1) != 0);
  }
  if (stamp && threadIdx.x == 0) stamp[1] = wall_clock64();
  const int want = __ockl_wgred_or_i32(has_src ? (1 << half) : 0);  // (a barrier: both tiles are complete behind it)
  if (stamp && threadIdx.x == 0) stamp[2] = wall_clock64();
  if (want && threadIdx.x == 0) *src_flag = serial;
  const int total = npair * G;
  const bool sampA = stat != nullptr && ((xA & 15) == 0 || q == 0);
  const bool sampB = stat != nullptr && xB != xA && (xB & 15) == 0;
  int n_farA = 0, n_farB = 0;
  for (int o = threadIdx.x; o < total; o += T) {
    const int p = o / G, gi = o - p * G;  // (G is a power of two)
    u32 a[4], c[4];
    if (want) {
      pk_scan16<G>(tA, tB, p, gi, npair, want, a, c);
    } else {
      a[0] = a[1] = a[2] = a[3] = c[0] = c[1] = c[2] = c[3] = pk_both(PK_INF);
    }
    if (sampA) n_farA += __popcll(__ballot(!(want & 1) || pk_hmin(pk_min(pk_min(a[0], a[1]), pk_min(a[2], a[3]))) > ESDF_FAR_D * ESDF_FAR_D));
    if (sampB) n_farB += __popcll(__ballot(!(want & 2) || pk_hmin(pk_min(pk_min(c[0], c[1]), pk_min(c[2], c[3]))) > ESDF_FAR_D * ESDF_FAR_D));
    uint4 r0, r1;  // y-rows 2p, 2p + 1: (slab A | slab B << 16) x 4 z
    const u32 mxa = pk_hmax(pk_max(pk_max(a[0], a[1]), pk_max(a[2], a[3]))), mxc = pk_hmax(pk_max(pk_max(c[0], c[1]), pk_max(c[2], c[3])));
    if (max(mxa, mxc) < PK_INF) {
      r0 = make_uint4(__builtin_amdgcn_perm(c[0], a[0], 0x05040100u), __builtin_amdgcn_perm(c[1], a[1], 0x05040100u),
                      __builtin_amdgcn_perm(c[2], a[2], 0x05040100u), __builtin_amdgcn_perm(c[3], a[3], 0x05040100u));
      r1 = make_uint4(__builtin_amdgcn_perm(c[0], a[0], 0x07060302u), __builtin_amdgcn_perm(c[1], a[1], 0x07060302u),
                      __builtin_amdgcn_perm(c[2], a[2], 0x07060302u), __builtin_amdgcn_perm(c[3], a[3], 0x07060302u));
    } else {
      // rare: outputs out of the 16-bit range (or slabs without a source): exact values from the tiles, 32 bits
      const int zc = 4 * gi, rA = 2 * p, rB = min(2 * p + 1, ylen - 1);
      u32 w0[4], w1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int z = zc0 + zc + k;
        const bool in_box = z >= b.lo[2] && z <= b.hi[2];
        u32 e[4];  // (slab A row 2p, slab A row 2p+1, slab B row 2p, slab B row 2p+1)
        e[0] = a[k] & 0xffffu, e[1] = a[k] >> 16, e[2] = c[k] & 0xffffu, e[3] = c[k] >> 16;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const bool slabB = v >= 2;
          const int row = (v & 1) ? rB : rA;
          if (!((want >> (slabB ? 1 : 0)) & 1))
            e[v] = in_box ? PK_NOSRC : 0u;
          else if (e[v] >= PK_INF) {
            const u32 ex = pk_slow_col(reinterpret_cast<const u32*>(slabB ? tB : tA), ZC, ylen, row, zc + k);
            u32* wat = wide + (long)(slabB ? xB : xA) * g.nyz + (long)(b.lo[1] + row) * g.nz + z;
            e[v] = pk2_encode(ex, wat, in_box);
          }
        }
        w0[k] = e[0] | (e[2] << 16), w1[k] = e[1] | (e[3] << 16);
      }
      r0 = make_uint4(w0[0], w0[1], w0[2], w0[3]);
      r1 = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    }
    // column tile of y-row y: tile0 + (y / RY) * tstride with RY = 8 >> gsh rows per tile (gsh: log2 of the tile's z width in
    // segments); inside its 128-byte row the lane's segment is (y % RY) << gsh + off + gi
    {
      const int y = 2 * p, ry1 = (8 >> gsh) - 1;
      store16(tmp16 + pk2_index(tile0 + (y >> (3 - gsh)) * tstride, q, ((y & ry1) << gsh) + off + gi, npx), r0);
    }
    if (2 * p + 1 < ylen) {
      const int y = 2 * p + 1, ry1 = (8 >> gsh) - 1;
      store16(tmp16 + pk2_index(tile0 + (y >> (3 - gsh)) * tstride, q, ((y & ry1) << gsh) + off + gi, npx), r1);
    }
  }
  if (stamp && threadIdx.x == 0) stamp[3] = wall_clock64();
  if ((sampA | sampB) && (threadIdx.x & 63) == 0) {
    if (sampA) {
      u32* sg = stat + 2 * ((xA >> 4) & (ESDF_NG - 1));
      atomicAdd(sg, (u32)n_farA);
      if (threadIdx.x == 0) atomicAdd(sg + 1, (u32)total);
    }
    if (sampB) {
      u32* sg = stat + 2 * ((xB >> 4) & (ESDF_NG - 1));
      atomicAdd(sg, (u32)n_farB);
      if (threadIdx.x == 0) atomicAdd(sg + 1, (u32)total);
    }
  }
}

// GMAX: widest chunk of the launch (the kernel's registers follow the widest body it contains: 110 VGPRs with the
// 8-segment body, 67 without)
template <int MODE, int GMAX, int NW>
__global__ void __launch_bounds__(1024)
k_esdf_zy_pk2(Geo g, Box3 b, const u64* __restrict__ infl, const u64* __restrict__ unk, uint4* __restrict__ tmp16,
              u32* __restrict__ wide, Pk2ZChunks ch, int z0a, int npx, int fastrow, u32* __restrict__ stat,
              u32* __restrict__ src_flag, u32 serial, unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nzc = ch.n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;  // slab pair q -> XCD q % 8 with all its chunk-blocks
  const int q = xcd + 8 * (slot / nzc);
  if (q >= npx) return;
  unsigned long long* stamp = dbg ? dbg + 8 * (size_t)blockIdx.x : nullptr;  // (FUELMI_ZY_TIMING)
  if (stamp && threadIdx.x == 0) stamp[0] = wall_clock64();
  const int c = slot % nzc;
  const int seg0 = ch.seg0[c], tile0 = ch.tile0[c], tstride = ch.tstride[c], gsh = ch.gsh[c], off = ch.off[c];
#define PK2_BODY(GG) zy_pk2_body<MODE, GG, NW>(g, b, infl, unk, tmp16, wide, z0a, seg0, tile0, tstride, gsh, off, npx, q, fastrow, stat, src_flag, serial, stamp, smem_raw)
  switch (ch.g[c]) {  // (uniform)
    case 8:
      if constexpr (GMAX >= 8) PK2_BODY(8);
      break;
    case 4:
      if constexpr (GMAX >= 4) PK2_BODY(4);
      break;
    case 2: PK2_BODY(2); break;
    default: PK2_BODY(1); break;
  }
#undef PK2_BODY
}

// exact 32-bit minimum of one x column from the 16-bit tile (PK_INF entries: the wide plane holds the value)
__device__ __noinline__ u32 x_slow_col16(const u32* tile, int c, const u32* __restrict__ wide_col, long row_stride, int xlen, int row) {
  u32 best = INF32;
  const int rmax = max(row, xlen - 1 - row);
  for (int r = 0; r <= rmax && (u32)__mul24(r, r) < best; ++r) {
    const u32 rr = (u32)__mul24(r, r);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int qrow = s ? row + r : row - r;
      if (qrow < 0 || qrow >= xlen || (s && !r)) continue;
      u32 v = (tile[(qrow >> 1) * 32 + c] >> (16 * (qrow & 1))) & 0xffffu;
      if (v == PK_NOSRC) continue;
      if (v >= PK_INF) v = wide_col[(long)qrow * row_stride];
      best = min(best, v + rr);
    }
  }
  return best;
}

template <int OUT>
__global__ void __launch_bounds__(512)
k_esdf_x_pk2(Geo g, Box3 b, const uint4* __restrict__ tmp16, const u32* __restrict__ wide, float* __restrict__ dist, Pk2Chunks ch,
             int z0a, int full8, u32* stat, volatile u32* h_stat, const u32* __restrict__ src_flag, u32 serial,
             unsigned long long* __restrict__ dbg) {
  constexpr int SEGS = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  forward_stat(stat, h_stat);
  uint4* tile = reinterpret_cast<uint4*>(smem_raw);  // [npx][SEGS] of uint4: halves = x-rows 2p, 2p + 1
  const int xlen = b.hi[0] - b.lo[0] + 1;
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int npx = (xlen + 1) >> 1;
  const int T = blockDim.x;
  const int total = npx * SEGS;
  const float resf = (float)g.res;
  unsigned long long* stamp = dbg ? dbg + 8 * (size_t)blockIdx.x : nullptr;  // (FUELMI_ZY_TIMING)
  if (stamp && threadIdx.x == 0) stamp[0] = wall_clock64();
  // Which tile (uniform).  The full-width tiles of y-row y run on XCD y % 8, one after the other: their 128-byte pieces of
  // the distance field are adjacent but not line-aligned (a y-row is nz * 4 bytes), so neighbours share lines -- on one
  // XCD the shared lines merge in its L2 instead of leaving as two partial writes.  Behind them (blocks >= full8) the
  // narrower pieces of the z range's remainder, 8 / w y-rows each.
  int gw = 8, gsh = 3, y0, seg0, t;
  if ((int)blockIdx.x < full8) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    y0 = xcd + 8 * (slot / ch.n8);
    if (y0 >= ylen) return;  // (the grid is padded to a multiple of 8 rows)
    const int c8 = slot % ch.n8;
    seg0 = 8 * c8;
    t = y0 * ch.n8 + c8;
  } else {
    t = (int)blockIdx.x - full8 + ylen * ch.n8;
    int c = ch.n8;
    while (c + 1 < ch.n && t >= ch.tile0[c + 1]) ++c;
    gw = ch.g[c], gsh = gw == 4 ? 2 : (gw == 2 ? 1 : 0);
    y0 = (t - ch.tile0[c]) * (8 >> gsh);
    seg0 = ch.seg0[c];
  }
  const bool any_src = *src_flag == serial;  // (no slab of the box holds a source: "no source" everywhere, nothing to scan)
  if (any_src) {
    // this tile's rows: one contiguous block
    const uint4* src = tmp16 + pk2_index(t, 0, 0, npx);
#pragma unroll 4
    for (int o = threadIdx.x; o < total; o += T) tile[o] = src[o];
  }
  if (stamp && threadIdx.x == 0) stamp[1] = wall_clock64();
  __syncthreads();
  if (stamp && threadIdx.x == 0) stamp[2] = wall_clock64();
  int n_far = 0;
  for (int o = threadIdx.x; o < total; o += T) {
    const int p = o >> 3, seg = o & 7;
    const int yy = y0 + (seg >> gsh);
    if (yy >= ylen) continue;
    uint4 ra, rb;
    pk_scan8<SEGS, false>(smem_raw, p, seg, npx, xlen, any_src, false, n_far, ra, rb);
    const int z = z0a + 4 * (seg0 + (seg & (gw - 1)));
    const long coloff = (long)(b.lo[1] + yy) * g.nz + z;
    if (any_src && max(max4(ra.x, ra.y, ra.z, ra.w), max4(rb.x, rb.y, rb.z, rb.w)) >= PK_INF) {
      // rare: outputs out of the 16-bit range -- exact from the tile and the wide plane
      u32* pa = &ra.x;
      u32* pb = &rb.x;
      const u32* wc = wide + (long)b.lo[0] * g.nyz + coloff;
      for (int k = 0; k < 4; ++k) {
        if (pa[k] >= PK_INF) pa[k] = x_slow_col16(reinterpret_cast<const u32*>(smem_raw), 4 * seg + k, wc + k, g.nyz, xlen, 2 * p);
        if (2 * p + 1 < xlen && pb[k] >= PK_INF)
          pb[k] = x_slow_col16(reinterpret_cast<const u32*>(smem_raw), 4 * seg + k, wc + k, g.nyz, xlen, 2 * p + 1);
      }
    }
    float* dst = dist + (long)(b.lo[0] + 2 * p) * g.nyz + coloff;
    x_store4<OUT>(dst, z, b.lo[2], b.hi[2], ra, resf);
    if (2 * p + 1 < xlen) x_store4<OUT>(dst + g.nyz, z, b.lo[2], b.hi[2], rb, resf);
  }
  if (stamp && threadIdx.x == 0) stamp[3] = wall_clock64();
}

// The column tiles of a box (x pass) and the chunks of its z/y pass.  Tiles: the aligned z range in pieces of 8, then 4,
// 2, 1 segments that add up to it exactly (100 voxels: 8 + 8 + 8 + 1 segments); a tile of width w holds 8 / w y-rows.
// Chunks: a tile piece wider than gz_max is cut into chunks of gz_max.  ESDF_NO_FIT beyond PK2_MAXCH pieces.
static int pk2_chunks(int ylen, int nseg, int gz_max, Pk2Chunks* xs, Pk2ZChunks* zs) {
  *xs = Pk2Chunks{};
  *zs = Pk2ZChunks{};
  const int n8 = nseg >> 3;  // full-width pieces: tile (y, c) = y * n8 + c
  int n = 0, nz = 0, seg = 0, tile = ylen * n8;
  auto add_z = [&](int seg_lo, int gw, int gsh, int tile0, int tstride) {
    const int gz = std::min(gw, gz_max);
    for (int o = 0; o < gw; o += gz) {
      if (nz == PK2_MAXZCH) return false;
      zs->seg0[nz] = (short)(seg_lo + o), zs->g[nz] = (signed char)gz, zs->gsh[nz] = (signed char)gsh, zs->off[nz] = (signed char)o;
      zs->tile0[nz] = tile0, zs->tstride[nz] = tstride;
      ++nz;
    }
    return true;
  };
  for (int c = 0; c < n8; ++c, ++n, seg += 8) {
    if (n == PK2_MAXCH) return ESDF_NO_FIT;
    xs->seg0[n] = (short)seg, xs->g[n] = 8, xs->tile0[n] = 0;  // (interleaved: see n8)
    if (!add_z(seg, 8, 3, c, n8)) return ESDF_NO_FIT;
  }
  xs->n8 = n8;
  while (seg < nseg) {  // the remainder in binary: pieces of 4, 2, 1 segments, tiles of 2, 4, 8 y-rows
    int gw = 4;
    while (gw > nseg - seg) gw >>= 1;
    const int gsh = gw == 4 ? 2 : (gw == 2 ? 1 : 0);
    if (n == PK2_MAXCH) return ESDF_NO_FIT;
    xs->seg0[n] = (short)seg, xs->g[n] = (short)gw, xs->tile0[n] = tile;
    if (!add_z(seg, gw, gsh, tile, 1)) return ESDF_NO_FIT;
    tile += (ylen + 8 / gw - 1) / (8 / gw);
    seg += gw;
    ++n;
  }
  xs->n = n;
  zs->n = nz;
  for (int k = n; k <= PK2_MAXCH; ++k) xs->tile0[k] = tile;
  for (int k = n; k < PK2_MAXCH; ++k) xs->g[k] = 1;
  if (n8 == n) xs->tile0[n8] = tile;
  return FUELMI_OK;
}
static u32 pk2_next_serial(fuelmi_map* m) {
  if (++m->esdf_serial == 0u) m->esdf_serial = 1u;
  return m->esdf_serial;
}
static u32* pk2_src_flag(fuelmi_map* m) { return m->esdf_stat + 2 * ESDF_NG + 2; }

template <int MODE, int GMAX, int NW>
static int launch_zy_pk2_g(fuelmi_map* m, const Box3& b, int z0a, int threads, size_t lds) {
  const int xlen = b.hi[0] - b.lo[0] + 1;
  const int npx = (xlen + 1) >> 1;
  const Pk2ZChunks& ch = m->pk2_zch;
  if (lds > 64 * 1024) HIPCHK(lds_attr_once(reinterpret_cast<const void*>(&k_esdf_zy_pk2<MODE, GMAX, NW>), 160 * 1024));
  const int grid = ((npx + 7) / 8) * 8 * ch.n;
  static const bool timing = getenv("FUELMI_ZY_TIMING") != nullptr;  // debug: where a workgroup's life goes
  unsigned long long* dbg = nullptr;
  if (timing) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dbg), (size_t)grid * 8 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(dbg, 0, (size_t)grid * 8 * sizeof(unsigned long long), m->stream));
  }
  STAGE_LAUNCH(m, (k_esdf_zy_pk2<MODE, GMAX, NW>), grid, threads, lds, m->g, b, (const u64*)m->infl_bits.p,
               (const u64*)m->unk_bits.p, reinterpret_cast<uint4*>(m->esdf_tmp16), m->esdf_tmp, ch, z0a, npx,
               (zy_fastrow() ? 1 : 0), MODE == 2 ? nullptr : esdf_stat_dev<0>(m), pk2_src_flag(m), m->esdf_serial, dbg);
  HIPCHK(hipGetLastError());
  if (timing) {
    char what[96];
    std::snprintf(what, sizeof what, "zy2-timing] GMAX %d chunks %d threads %d lds %zu", GMAX, (int)ch.n, threads, lds);
    { const int rc_ = pass_timing_report(m, dbg, grid, what); if (rc_ != FUELMI_OK) return rc_; }
  }
  return FUELMI_OK;
}
// the packed family's z/y pass: ESDF_NO_FIT when the box is not its kind (z extents above 255 voxels, nz % 4 != 0,
// y lines too long for two tiles, x lines too long for the x pass's tile)
template <int MODE>
static int launch_zy_pk2(fuelmi_map* m, const Box3& b) {
  const Geo& g = m->g;
  const int xlen = b.hi[0] - b.lo[0] + 1, ylen = b.hi[1] - b.lo[1] + 1, zlen = b.hi[2] - b.lo[2] + 1;
  if ((g.nz % 4) != 0 || zlen > 255 || !m->esdf_tmp16) return ESDF_NO_FIT;
  const int z0a = b.lo[2] & ~3, z1a = b.hi[2] | 3;
  const int zlen_a = z1a - z0a + 1;
  const int npair = (ylen + 1) >> 1;
  if ((size_t)((xlen + 1) >> 1) * 128 > 150 * 1024) return ESDF_NO_FIT;  // (the x pass's tile)
  // widest chunk: 4 segments (16 voxels; the 8-segment body needs 110 VGPRs and measured 46 against 34 us on the
  // 400^2 x 100 map) while the two tiles of a slab pair stay within 52 KB (three workgroups per CU), else 2
  int g_max = 4;
  while (g_max > 2 && (size_t)npair * g_max * 32 > 52 * 1024) g_max >>= 1;
  const size_t lds = (size_t)npair * g_max * 32;
  if (lds > 160 * 1024 - 64) return ESDF_NO_FIT;
  if (pk2_chunks(ylen, zlen_a >> 2, g_max, &m->pk2_ch, &m->pk2_zch) != FUELMI_OK) return ESDF_NO_FIT;
  if ((size_t)m->pk2_ch.tile0[m->pk2_ch.n] * (size_t)((xlen + 1) >> 1) * 128 > m->esdf_tmp16_bytes) return ESDF_NO_FIT;
  // a half fills its npair rows in two trips: 128 / 256 lanes for 400- / 800-voxel y lines (measured: 33.5 against 35.5 us
  // with one trip on the 400^2 x 100 map, 158 against 203 us on 800^2 x 200, whose 896-thread workgroups fit one per CU)
  const int threads = 2 * std::min(256, std::max(64, ((npair / 2 + 63) / 64) * 64));
  const bool wide = zlen_a > 128;  // aligned z-lines of up to 128 / 256 bits
  (void)pk2_next_serial(m);
  if (g_max == 8) return wide ? launch_zy_pk2_g<MODE, 8, 4>(m, b, z0a, threads, lds) : launch_zy_pk2_g<MODE, 8, 2>(m, b, z0a, threads, lds);
  if (g_max == 4) return wide ? launch_zy_pk2_g<MODE, 4, 4>(m, b, z0a, threads, lds) : launch_zy_pk2_g<MODE, 4, 2>(m, b, z0a, threads, lds);
  return wide ? launch_zy_pk2_g<MODE, 2, 4>(m, b, z0a, threads, lds) : launch_zy_pk2_g<MODE, 2, 2>(m, b, z0a, threads, lds);
}

// the x pass behind launch_zy_pk2 (same box, same serial, same chunks)
template <int OUT>
static int launch_x_pk2(fuelmi_map* m, const Box3& b) {
  const Geo& g = m->g;
  const int xlen = b.hi[0] - b.lo[0] + 1;
  const int z0a = b.lo[2] & ~3;
  const int npx = (xlen + 1) >> 1;
  const size_t lds = (size_t)npx * 8 * 16;
  if (lds > 64 * 1024) HIPCHK(lds_attr_once(reinterpret_cast<const void*>(&k_esdf_x_pk2<OUT>), 150 * 1024));
  const int threads = lds > 32 * 1024 ? 512 : 256;
  static const bool timing = getenv("FUELMI_ZY_TIMING") != nullptr;
  const int ylen = b.hi[1] - b.lo[1] + 1;
  const int full8 = ((ylen + 7) / 8) * 8 * m->pk2_ch.n8;  // blocks of the full-width tiles, y-rows padded to a multiple of 8
  const int grid = full8 + (m->pk2_ch.tile0[m->pk2_ch.n] - ylen * m->pk2_ch.n8);
  unsigned long long* dbg = nullptr;
  if (timing) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dbg), (size_t)grid * 8 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(dbg, 0, (size_t)grid * 8 * sizeof(unsigned long long), m->stream));
  }
  STAGE_LAUNCH(m, (k_esdf_x_pk2<OUT>), grid, threads, lds, g, b, reinterpret_cast<const uint4*>(m->esdf_tmp16),
               (const u32*)m->esdf_tmp, m->dist, m->pk2_ch, z0a, full8, esdf_stat_dev<OUT>(m), esdf_stat_host(m),
               (const u32*)pk2_src_flag(m), m->esdf_serial, dbg);
  HIPCHK(hipGetLastError());
  if (timing) {
    char what[96];
    std::snprintf(what, sizeof what, "x2-timing] threads %d lds %zu", threads, lds);
    { const int rc_ = pass_timing_report(m, dbg, grid, what); if (rc_ != FUELMI_OK) return rc_; }
  }
  return FUELMI_OK;
}

template <int OUT, bool FAR>
static int launch_x4(fuelmi_map* m, const Box3& b) {
  const int xlen = b.hi[0] - b.lo[0] + 1;
  // the 32-column tile is faster whenever it fits (measured on 800-voxel lines: 0.32 vs 0.37 ms), the
  // 16-column one extends the vector path to x lines of up to 2400 voxels
  const bool narrow = (size_t)xlen * 128 > 150 * 1024;
  const int rc = narrow ? launch_x4s<OUT, 4, FAR>(m, b) : launch_x4s<OUT, 8, FAR>(m, b);
  if (rc != ESDF_NO_FIT || !FAR) return rc;
  return narrow ? launch_x4s<OUT, 4, false>(m, b) : launch_x4s<OUT, 8, false>(m, b);
}

template <int MODE, bool FAR>
static int launch_zy(fuelmi_map* m, const Box3& b) {
  if (use_vec4(m->g, b.hi[0] - b.lo[0] + 1)) {
    const int rc = launch_zy4<MODE, FAR>(m, b);
    return (FAR && rc == ESDF_NO_FIT) ? launch_zy4<MODE, false>(m, b) : rc;
  }
  const Geo& g = m->g;
  const int xlen = b.hi[0] - b.lo[0] + 1, ylen = b.hi[1] - b.lo[1] + 1, zlen = b.hi[2] - b.lo[2] + 1;
  // z-chunk: LDS tile <= 16 KiB so several WGs share a CU, chunks balanced over the z extent
  int zc_max = std::max(1, (16 * 1024) / (2 * ylen));
  if (zc_max > zlen) zc_max = zlen;
  if (zc_max > 256) zc_max = 256;
  int nzc = (zlen + zc_max - 1) / zc_max;
  int ZC = (zlen + nzc - 1) / nzc;
  size_t lds = (size_t)ylen * ZC * sizeof(unsigned short);
  if (lds > 160 * 1024) {
    fuelmi_set_error("ESDF y-line of %d voxels does not fit the LDS tile", ylen);
    return FUELMI_ELIMIT;
  }
  if (lds > 64 * 1024)
    HIPCHK(lds_attr_once(reinterpret_cast<const void*>(&k_esdf_zy<MODE>), 160 * 1024));
  STAGE_LAUNCH(m, (k_esdf_zy<MODE>), xlen * nzc, 256, lds, g, b, (const u64*)m->infl_bits.p, (const u64*)m->unk_bits.p,
               m->esdf_tmp, ZC, nzc);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

template <int S, int OUT>
static int launch_x_s(fuelmi_map* m, const Box3& b) {
  const Geo& g = m->g;
  const int xlen = b.hi[0] - b.lo[0] + 1, ylen = b.hi[1] - b.lo[1] + 1, zlen = b.hi[2] - b.lo[2] + 1;
  size_t lds = (size_t)xlen * S * sizeof(u32);
  if (lds > 64 * 1024)
    HIPCHK(lds_attr_once(reinterpret_cast<const void*>(&k_esdf_x<S, OUT>), 160 * 1024));
  int ncol = ylen * zlen;
  STAGE_LAUNCH(m, (k_esdf_x<S, OUT>), (ncol + S - 1) / S, 256, lds, g, b, (const u32*)m->esdf_tmp, m->dist);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

template <int OUT, bool FAR>
static int launch_x(fuelmi_map* m, const Box3& b) {
  const int xlen = b.hi[0] - b.lo[0] + 1;
  if (use_vec4(m->g, xlen)) return launch_x4<OUT, FAR>(m, b);
  const size_t budget = 64 * 1024;
  if ((size_t)xlen * 32 * 4 <= budget) return launch_x_s<32, OUT>(m, b);
  if ((size_t)xlen * 16 * 4 <= 128 * 1024) return launch_x_s<16, OUT>(m, b);
  if ((size_t)xlen * 8 * 4 <= 128 * 1024) return launch_x_s<8, OUT>(m, b);
  fuelmi_set_error("ESDF x-line of %d voxels does not fit the LDS tile", xlen);
  return FUELMI_ELIMIT;
}

// Which kernels: the y pass of every update counts, on one slab of every group of 16, the outputs further than
// ESDF_FAR_D voxels from all sources; the x pass hands the counts to the host.  The host keeps the latest pair of
// every group -- a statistic per PLACE, not "of the previous update" -- and an update runs the FAR kernels (block /
// line minima bound the scan) when most outputs of the slabs it covers were far the last time they were updated:
// an explored hall, optimistic_ maps.  A local bound that alternates between a hall and a fresh frustum therefore
// gets the right kernels for both from the second visit on; places never seen follow the last decision.  All
// families are exact; fuelmi_map_set_esdf_family pins one (tests, A/B runs).
static bool esdf_use_far(fuelmi_map* m, const Box3& b) {
  const volatile unsigned long long* h = reinterpret_cast<const volatile unsigned long long*>(esdf_stat_host(m));
  // entries are self-describing (forward_stat): take whatever is new -- of the groups THIS box covers (a full-width box
  // of a 400-voxel map: 25 words, not all 256; the other places are looked at when a box reaches them)
  const int g_lo = b.lo[0] >> 4, g_n = std::min((b.hi[0] >> 4) - g_lo + 1, ESDF_NG);
  for (int gi = 0; gi < g_n; ++gi) {
    const int g = (g_lo + gi) & (ESDF_NG - 1);
    const unsigned long long v = h[g];
    const unsigned short tag = (unsigned short)(v >> 48);
    if (tag != 0 && tag != m->far_tag[g]) {
      m->far_tag[g] = tag;
      m->far_hist[g][0] = (u32)(v & 0xffffffu);
      m->far_hist[g][1] = (u32)((v >> 24) & 0xffffffu);
    }
  }
  unsigned long long nf = 0, nt = 0;
  for (int g = b.lo[0] >> 4; g <= (b.hi[0] >> 4); ++g) nf += m->far_hist[g & (ESDF_NG - 1)][0], nt += m->far_hist[g & (ESDF_NG - 1)][1];
  if (nt != 0) m->far_last = 2 * nf > nt;
  static const bool dbg = getenv("FUELMI_ESDF_DEBUG") != nullptr;
  if (dbg) std::fprintf(stderr, "[fuelmi] esdf regime: far %llu of %llu sampled outputs -> %s kernels\n", nf, nt,
                        m->far_last ? "far-field" : "plain");
  return m->far_last;
}

template <int MODE>
static int launch_zy_family(fuelmi_map* m, const Box3& b, int fam, int* ran) {
  if (fam == FUELMI_ESDF_PLAIN) {
    const int rc = launch_zy_pk2<MODE>(m, b);
    m->esdf_pk2_last = rc != ESDF_NO_FIT;
    if (rc != ESDF_NO_FIT) {
      *ran = FUELMI_ESDF_PLAIN;
      return rc;
    }
    fam = FUELMI_ESDF_PLAIN32;
  }
  *ran = fam;
  if constexpr (MODE != 2) {  // (the negative pass of signed maps never runs the far-field kernels)
    if (fam == FUELMI_ESDF_FAR) return launch_zy<MODE, true>(m, b);
  }
  *ran = FUELMI_ESDF_PLAIN32;
  return launch_zy<MODE, false>(m, b);
}

int esdf_update(fuelmi_map* m) {
  const Box3& b = m->local_bound;
  const int fam = m->esdf_family_pin >= 0 ? m->esdf_family_pin : (esdf_use_far(m, b) ? FUELMI_ESDF_FAR : FUELMI_ESDF_PLAIN);
  const bool far = fam == FUELMI_ESDF_FAR;
  int rc, ran = fam;
  {
    StageScope sc(m, FUELMI_K_ESDF_ZY, nullptr, true);
    rc = m->cfg.optimistic ? launch_zy_family<1>(m, b, fam, &ran) : launch_zy_family<0>(m, b, fam, &ran);
  }
  if (rc) return rc;
  m->esdf_family_last = ran;
  {
    StageScope sc(m, FUELMI_K_ESDF_X, nullptr, true);
    rc = ESDF_NO_FIT;
    if (ran == FUELMI_ESDF_PLAIN && m->esdf_pk2_last)
      rc = launch_x_pk2<0>(m, b);  // the packed family: both passes on 16-bit lanes, 16-bit tile-contiguous hand-over
    if (rc == ESDF_NO_FIT) rc = far ? launch_x<0, true>(m, b) : launch_x<0, false>(m, b);
  }
  if (rc) return rc;
  if (m->cfg.signed_dist) {  // inside obstacles the nearest free voxel is never far: plain kernels
    {
      StageScope sc(m, FUELMI_K_ESDF_ZY, nullptr, true);
      int ran2;
      rc = launch_zy_family<2>(m, b, far ? FUELMI_ESDF_PLAIN : fam, &ran2);
      if (ran2 != FUELMI_ESDF_PLAIN) m->esdf_pk2_last = false;
    }
    if (rc) return rc;
    StageScope sc(m, FUELMI_K_ESDF_X, nullptr, true);
    rc = m->esdf_pk2_last ? launch_x_pk2<1>(m, b) : launch_x<1, false>(m, b);
  }
  return rc;
}

extern "C" int fuelmi_map_set_esdf_family(fuelmi_map* m, int family) {
  ARGCHK(m && family >= FUELMI_ESDF_AUTO && family <= FUELMI_ESDF_PLAIN32);
  m->esdf_family_pin = family;
  return FUELMI_OK;
}
extern "C" int fuelmi_map_last_esdf_family(const fuelmi_map* m) {
  ARGCHK(m);
  return m->esdf_family_last;
}
