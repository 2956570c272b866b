// frontier_internal.h -- structures shared by frontier.hip (scan + clustering) and
// frontier_split.hip (splitLargeFrontiers / down-sampling on the device).
#ifndef FUELMI_FRONTIER_INTERNAL_H_
#define FUELMI_FRONTIER_INTERNAL_H_

#include <algorithm>
#include <cstring>
#include <list>
#include <memory>
#include <vector>

#include "fuelmi_internal.h"

#define NOCLAIM 0xFFFFFFFFu
#define SZ_CH 1024

struct KeptRec {  // one per kept cluster, in rank order (ascending claimer address)
  u32 addr, slot, size, off;  // off: first position of the cluster in the grouped cell array
  unsigned long long sum[3];  // voxel-index sums / AABB of cells outside their chunk's leading key
  u32 box[6];
  u32 pad[2];
};

// the per-search part of the arguments lives in device memory (refreshed by k_load_var from a pinned
// host copy), so that the kernel chain can be replayed as a hipGraph with constant kernel arguments
struct FVar {
  Box3 sbox;   // scanned index box, inclusive
  int w0;      // first word processed (multiple of 256)
  int nwords;  // words processed (multiple of 256)
  int nblocks; // nwords / 256
  int fresh;   // 1: frontier_flag_ counts as all-zero and is rewritten by this search (fuelmi_frontier_reset
               // folded into the first kernel; the processed words then cover every flag that can be set)
  // Q0 cells can only exist inside the region touched since the last search (scan box, boxes of the
  // clusters just dropped; the whole exploration box after untracked changes): the CCL tiles cover
  // this part of F.qbox only
  Box3 qreg;
  int nty, ntiles;
  u32 epoch;  // search counter, echoed next to the result the host polls for
  int pad2;
  int nty_f, ntiles_f;  // tile grid of the fast chain (its tiles are sized independently of the legacy ones)
  int ftx, fty;         // ... and its tile, chosen per search: few big tiles would serialise a small region
  // the fast chain's tiles cover the x/y bounding rectangle of qreg and sbox (NQ seeds sit on the box_max face,
  // one voxel outside the Q box), all z
  int px0, py0, px1, py1;
  int ntx_f;
  int pad3[3];
};

// ---- fast path of the clustering chain (frontier.hip, "tile-root resolve") ---------------------------------
// A workgroup labels one spatial tile in LDS (k_tile_ccl) and describes every tile-local component by ONE record;
// the cross-tile merge, the claims, the cluster sizes, the kept list and its ranking then run on those few
// thousand records inside a single workgroup's LDS (k_resolve) instead of on the cells through global atomics.
// No compaction of the cells in front of it: a tile finds its cells in the Q0 bit-plane, a cell's component is
// looked up by voxel address (vlab), and the grouped output positions come from per-(component, x-row) counts.
#define FR_RCAP 8192            // tile-local components per search (8 per-XCD ranges of FR_RC8)
#define FR_RC8 (FR_RCAP / 8)
#define FR_TCELL 2048           // Q0 cells per tile
#define FR_TPAIR 1536           // touching pairs of runs a tile lists (beyond: union-find on the spot)
#define FR_TROOT 128            // components per tile (they are numbered in one byte; 0xFF is free)
#define FR_TXS 16               // x-rows per tile at most (stride of the per-row counts)
#define FR_KCAP 256             // kept clusters
#define FR_PCAP (1u << 18)      // cross-tile adjacency records
// The chain's counters (fctr): 32 logical words -- [0..7] roots per XCD range, [9] capacity code, [16..23] pairs per XCD --
// each on a 128-byte line of its own, and the last-workgroup-done counter of k_tile_cross on a 33rd.  They are hit by
// returning memory-side atomics of hundreds of workgroups: atomics to ONE line queue up behind each other whatever word
// they address (round 6: with the done counter on the pair counters' line the busiest tile's cross phase went 13 -> 19 us).
#define FCTR(k) ((k) * 32)
#define FR_DONE_CTR FCTR(32)
#define FR_NCTR (41 * 32)  // (FR_DONE_CTR: the global count, FR_DONE_CTR + FCTR(1 + xcd): per XCD)
#define FR_PMCAP 16384          // entries of the (kept cluster x tile column) matrix k_resolve scans in its LDS
#define FR_REFORDER_AUTO 26624u   // cfg.reference_order == 2: searches whose clusters all hold at most this many cells use the
                                  // reference's order (what frontier_order.hip sweeps inside LDS)
#define FR_UNCLAIMED 0xFFFFFFFFu  // rcode: component claimed by nobody (no flag, no cluster)
#define FR_NOTKEPT 0xFFFFFFFEu    // rcode: claimed (flag set) but its cluster is too small
struct TRec {  // one tile-local component
  u32 size;            // cells
  u32 sx, sy, sz;      // voxel-index sums
  u32 lo[3], hi[3];    // index box
  u32 tx;              // tile column it lives in
  u32 own;             // lowest address among its own cells inside the scan box (NOCLAIM: none)
};

struct FArgs {
  Box3 qbox;  // Q0 index box (isInBox & z >= iz_min), inclusive
  const FVar* var;
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
  // grouping of the kept cells by cluster rank (stable 8-bit radix multisplit)
  int* slot2rank;    // [cap_q + cap_s] valid only at kept slots
  struct KeptRec* krec;  // [cap_kept] cluster records in rank order
  u32* ms_key[2];    // [cap_q]
  u32* ms_val[2];    // [cap_q]
  u32* ms_hist;      // [256][ms_nb_max]
  u32 ms_nb_max;
  u32* info_part;                // [cap_q / SZ_CH + 1][10] per-chunk records (key, sums, min, max)
  // result staging in pinned HOST memory, written by the last kernels themselves (no blit copies):
  u32* h_counts;          // [16]
  struct KeptRec* h_rec;  // [cap_kept]
  u32* h_part;            // like info_part
  u32* h_cells;           // [cap_q] grouped cell addresses
  u32 hcells_direct_max;  // the fast chain writes h_cells itself only for searches of at most this many Q0 cells;
                          // larger lists stay on the device and are fetched when somebody asks (frontier_cells_ready)
  int keys_from_slots;    // multisplit pass 0 derives (key, value) from cell_slot / slot2rank / cell_adr
  // fast path
  FVar* var_w;            // == var (the first kernel refreshes it from the pinned host copy)
  u32* tq;                // [tiles][items] Q0 bits of every 32-voxel segment of every tile (segment = line * nseg + c)
  u32* ts;                // [tiles][items] NQ seed bits, same layout
  unsigned char* vlab;    // [N] tile-local component number of every Q0 cell of the search (valid where qb is set)
  unsigned char* tlab;    // [tiles][FR_TCELL] the same by tile-local cell index (coalesced source of k_tile_out)
  u32* t_base;            // [tiles] id of the tile's component 0 (XCD range | index)
  u32* t_nroots;          // [tiles] components of the tile
  struct TRec* trec;      // [FR_RCAP]
  u32* tclaim;            // [FR_RCAP] lowest claimer address of every tile root (own in-box cells, NQ seeds)
  u32* rcode;             // [FR_RCAP] cluster rank | FR_NOTKEPT | FR_UNCLAIMED per tile root
  unsigned short* rrow;   // [FR_RCAP][FR_TXS] cells of the tile root in each x-row of its tile
  u32* pm;                // [FR_KCAP][pm_stride] cells of kept cluster (by rank) in the tile columns in front of column tx
  int pm_stride;
  u32* pairs;             // [8][FR_PCAP / 8] distinct pairs of tile roots that touch across a tile face (lo << 16 | hi)
  u32* fctr;              // [FR_NCTR] the chain's counters, one per cache line: FCTR(k), FR_DONE_CTR (above)
  int fast;               // the result of the last search came from the fast path
  unsigned long long* dbg;  // FUELMI_FR_TIMING: per-block phase time stamps of the fast chain's kernels (else null)
};
// Result words in pinned host memory: the data stores of a workgroup, a barrier, then ONE lane stores the stamp
// the polling host waits for with RELEASE semantics at system scope.  (A relaxed stamp was tried -- the release
// costs nothing measurable in these kernels -- and made a parity test flaky: stores of different waves leave
// through different L2 channels, nothing orders them against the stamp without the release.)
#define FR_HCELLS_DIRECT 32768u  // (see FArgs::hcells_direct_max)
#define FR_DBG_SLOTS 16
#define FR_DBG_MARK(F, blk, k)                                                                   \
  do {                                                                                           \
    if ((F).dbg && threadIdx.x == 0) (F).dbg[(size_t)(blk) * FR_DBG_SLOTS + (k)] = wall_clock64(); \
  } while (0)

#ifdef __HIPCC__
// Relaxed accesses at AGENT scope: the store goes through the XCD's L2 to memory, the load comes from there -- how
// workgroups of ONE launch (which may sit on different XCDs, each with its own write-back L2) hand data to each other
// without a release fence (= an L2 write-back, which also flushes what the ESDF passes of the same cycle hold dirty).
// Order: data stores, s_waitcnt vmcnt(0), workgroup barrier, then the counter.
__device__ __forceinline__ void st_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_vm_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// two tile-root records (48 bytes each, 16-byte aligned) with six 16-byte agent-scope loads and ONE wait
__device__ __forceinline__ void ld_agent_trec2(const TRec* p0, const TRec* p1, TRec& A, TRec& B) {
  static_assert(sizeof(TRec) == 48, "TRec is fetched as three 16-byte pieces");
  uint4 a0, a1, a2, b0, b1, b2;
  asm volatile(
      "global_load_dwordx4 %0, %6, off sc1\n\t"
      "global_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %6, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\t"
      "global_load_dwordx4 %4, %7, off offset:16 sc1\n\t"
      "global_load_dwordx4 %5, %7, off offset:32 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(b0), "=&v"(b1), "=&v"(b2)
      : "v"(p0), "v"(p1)
      : "memory");
  A.size = a0.x, A.sx = a0.y, A.sy = a0.z, A.sz = a0.w, A.lo[0] = a1.x, A.lo[1] = a1.y, A.lo[2] = a1.z, A.hi[0] = a1.w;
  A.hi[1] = a2.x, A.hi[2] = a2.y, A.tx = a2.z, A.own = a2.w;
  B.size = b0.x, B.sx = b0.y, B.sy = b0.z, B.sz = b0.w, B.lo[0] = b1.x, B.lo[1] = b1.y, B.lo[2] = b1.z, B.hi[0] = b1.w;
  B.hi[1] = b2.x, B.hi[2] = b2.y, B.tx = b2.z, B.own = b2.w;
}
// compact index of the Q0 cell / NQ seed at voxel address a (valid after k_pred + k_scan_sums of this search)
__device__ __forceinline__ u32 rank_q(const FArgs& F, long a) {
  int w = (int)(a >> 6);
  int rel = w - F.var->w0;
  u64 pk = F.blockscan[rel >> 8] + F.pref[rel];
  return (u32)pk + (u32)__popcll(F.qb[w] & ((1ull << (a & 63)) - 1ull));
}
// Q0 bits of z-line (xx, yy) at z = 32 c - 1 .. 32 c + 32 (bit j <-> z = 32 c - 1 + j), from the tile that owns the
// line (any tile of the search); 0 outside the tiled rectangle
// 64-lane reductions on the DPP data path (quads, half rows, rows, then the two row broadcasts of GFX9; the total is
// read from lane 63 and returned to every lane).  A __shfl_xor butterfly is six trips through the LDS crossbar per
// value; the record phases reduce 7 to 11 values at a time.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_mov(u32 old, u32 v) {
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
#define FUELMI_WAVE_RED(NAME, OP, ID)                                                        \
  __device__ __forceinline__ u32 NAME(u32 v) {                                               \
    v = OP(v, dpp_mov<0xB1, 0xF>(ID, v));  /* quad_perm [1,0,3,2] */                         \
    v = OP(v, dpp_mov<0x4E, 0xF>(ID, v));  /* quad_perm [2,3,0,1] */                         \
    v = OP(v, dpp_mov<0x141, 0xF>(ID, v)); /* row_half_mirror */                             \
    v = OP(v, dpp_mov<0x140, 0xF>(ID, v)); /* row_mirror: every lane holds its row's value */ \
    v = OP(v, dpp_mov<0x142, 0xA>(ID, v)); /* row_bcast15 into rows 1 and 3 */               \
    v = OP(v, dpp_mov<0x143, 0xC>(ID, v)); /* row_bcast31 into rows 2 and 3 */               \
    return (u32)__builtin_amdgcn_readlane((int)v, 63);                                       \
  }
__device__ __forceinline__ u32 red_add_(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 red_min_(u32 a, u32 b) { return min(a, b); }
__device__ __forceinline__ u32 red_max_(u32 a, u32 b) { return max(a, b); }
FUELMI_WAVE_RED(wave_add_u32, red_add_, 0u)
FUELMI_WAVE_RED(wave_min_u32, red_min_, 0xFFFFFFFFu)
FUELMI_WAVE_RED(wave_max_u32, red_max_, 0u)

// (branch-free: a load inside a branch is waited for inside it, and the callers issue several of these side by side --
// coordinates outside the rectangle are clamped into it and the result masked, the segments before the first / behind
// the last of a line read the line's own word again)
__device__ __forceinline__ u64 q_window34(const Geo& g, const FVar& V, const FArgs& F, int xx, int yy, int c) {
  const bool inside = xx >= V.px0 && xx <= V.px1 && yy >= V.py0 && yy <= V.py1;
  const int xc = min(max(xx, V.px0), V.px1), yc = min(max(yy, V.py0), V.py1);
  const int nseg = (g.nz + 31) >> 5;
  const int tx = (xc - V.px0) / V.ftx, ty = (yc - V.py0) / V.fty;
  const int lx = xc - V.px0 - tx * V.ftx, ly = yc - V.py0 - ty * V.fty;
  const u32* p = F.tq + (size_t)(tx * V.nty_f + ty) * (size_t)(V.ftx * V.fty * nseg) + (size_t)(lx * V.fty + ly) * nseg + c;
  const bool has_lo = c > 0, has_hi = c + 1 < nseg;
  const u32 lo = p[has_lo ? -1 : 0], mid = p[0], hi = p[has_hi ? 1 : 0];
  const u64 w = (u64)(has_lo ? lo >> 31 : 0u) | ((u64)mid << 1) | ((u64)(has_hi ? hi & 1u : 0u) << 33);
  return inside ? w : 0ull;
}

// ... and the three voxels z - 1, z, z + 1 of that line (bit 0 <-> z - 1)
__device__ __forceinline__ u32 q_bits3(const Geo& g, const FVar& V, const FArgs& F, int xx, int yy, int z) {
  const int c = z >> 5;
  return (u32)(q_window34(g, V, F, xx, yy, c) >> (z - 32 * c)) & 7u;
}
__device__ __forceinline__ u32 rank_s(const FArgs& F, long a) {
  int w = (int)(a >> 6);
  int rel = w - F.var->w0;
  u64 pk = F.blockscan[rel >> 8] + F.pref[rel];
  return (u32)(pk >> 32) + (u32)__popcll(F.sb[w] & ((1ull << (a & 63)) - 1ull));
}
#endif

#define MS_CH 2048  // cells per multisplit block (256 threads x 8)
#define NOKEY 0xFFFFFFFFu


// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct HViewpoint {  // Viewpoint (frontier_finder.h:25-33)
  double pos[3];
  double yaw;
  int visib_num;
};
struct HCluster {
  std::vector<HViewpoint> viewpoints;
  std::vector<int> cells;  // ascending voxel addresses
  double avg[3], bmin[3], bmax[3];
  // A freshly found cluster (tmp_frontiers_) still lives in the pinned result buffer: its cell list is
  // copied out only when somebody keeps it (commit) -- a 140 k-cell surface costs ~25 us to copy,
  // 10 % of a plan cycle, and the buffer stays valid until the next search.
  std::vector<float> filtered;  // filtered_cells_ (xyz triples), only for clusters found with cfg.split
  // committed clusters also keep their cells in the finder's device pool (remove_changed and
  // isFrontierCovered test them there instead of re-uploading them on every search)
  size_t pool_off = (size_t)-1;
  const int* lazy = nullptr;
  u32 lazy_n = 0;
  int lazy_seed = -1;  // NQ seed address to merge in, or -1
  // A committed cluster whose pinned result buffer has been handed to the next search keeps its cells in the finder's
  // device pool only (dev_n cells at pool_off, an NQ seed last): nothing on the host until somebody asks
  // (frontier_fetch_cluster).  Copying every kept list out before each search was ~30 us of every streaming frame.
  bool dev_only = false;
  u32 dev_n = 0;
  size_t size() const { return dev_only ? (size_t)dev_n : (lazy ? (size_t)lazy_n + (lazy_seed >= 0 ? 1u : 0u) : cells.size()); }
  void copy_to(int* out) const {  // ascending addresses
    if (!lazy) {
      if (!cells.empty()) memcpy(out, cells.data(), cells.size() * sizeof(int));
      return;
    }
    if (lazy_seed < 0) {
      if (lazy_n) memcpy(out, lazy, (size_t)lazy_n * sizeof(int));
      return;
    }
    const size_t k = (size_t)(std::lower_bound(lazy, lazy + lazy_n, lazy_seed) - lazy);
    if (k) memcpy(out, lazy, k * sizeof(int));
    out[k] = lazy_seed;
    if (lazy_n > k) memcpy(out + k + 1, lazy + k, (lazy_n - k) * sizeof(int));
  }
  void materialize() {
    if (!lazy) return;
    std::vector<int> v(size());
    copy_to(v.data());
    cells.swap(v);
    lazy = nullptr;
  }
};

struct fuelmi_frontier {
  fuelmi_map* map = nullptr;  // cleared if the map is destroyed first (then only _destroy is legal)
  int device = 0;
  fuelmi_frontier_cfg cfg;
  int iz_min = 0;
  Plane flag, qb, sb;
  Plane flag2;  // the spare flag plane (all-zero or being zeroed): a reset swaps the two
  int flag_cur = 0;
  hipEvent_t ev_tail = nullptr;
  bool zero_deferred = false;  // the retired plane's zeroing is still to be queued (frontier_finish_reset)
  FArgs F;
  // A fresh search (fuelmi_frontier_reset) swaps to the OTHER flag plane -- and, with it, to the other set of every
  // per-search buffer (F2: tile tables, records, grouped cells, pinned result block, per-search variables) and the
  // other stream: the tail of the previous search (k_tile_out: flags + grouped cell lists, ~15-30 us) then runs BESIDE
  // the next search's chain instead of in front of it.  Incremental searches (no reset) stay on one plane, one
  // buffer set, one stream: their flags are a true dependency.
  FArgs F2;
  hipStream_t stream2 = nullptr;
  struct ScratchRec {  // a device buffer of F (byte offset of its pointer inside FArgs, size): F2 gets a twin
    size_t field_off, bytes;
  };
  std::vector<ScratchRec> f_scratch;
  void* h_pin2 = nullptr;
  bool pool_dirty = false;  // the cell pool was written since the last plane swap (the new stream then waits for the old one)
  std::vector<void*> allocs;
  std::list<HCluster> frontiers, dormant, tmp;
  // which = 3: the new clusters of the search BEFORE the last fuelmi_frontier_reset.  Their cell lists sit in the
  // retired buffer set (see F2), which nothing touches until the next reset -- a caller that works in cycles reads
  // cycle k - 1's cells while cycle k runs on the device (fuelmi_bench_cycles_delivered does).
  std::list<HCluster> prev;
  hipEvent_t ev_planes_read = nullptr;  // behind the last kernel of the running search that reads the map's occupancy planes
  hipEvent_t* tl_ev = nullptr;          // FUELMI_STREAM_TIMING=2: four timing events of the current frame's chain (start,
                                        // planes read, resolved, tail done), recorded by frontier_enqueue_fast
  hipEvent_t ev_prev = nullptr;  // the retired search's tail (and the copy of its grouped cells to the host) have completed
  bool prev_pending = false;
  unsigned long long fusion_at_begin = 0;  // map->fusion_count when the running search began
  bool keep_prev = false;  // fuelmi_frontier_keep_previous
  std::vector<int> removed_ids;
  hipStream_t stream = nullptr;  // frontier work runs beside the map's own stream
  hipEvent_t ev_dep = nullptr;
  unsigned long long planes_waited[2] = {0, 0};  // fuelmi_map::planes_ver + 1 each search stream (by flag plane) has waited for
  bool mark_planes_read = false;  // record ev_planes_read between the chain's first two kernels (see fuelmi_map::late_readers)
  void* d_stage = nullptr;
  size_t d_stage_bytes = 0;
  void* h_pin = nullptr;  // pinned result staging
  size_t pin_bytes = 0;
  int last_nkept = 0, nb_launch = 0, npass = 1;
  FVar* h_var = nullptr;  // pinned per-search arguments
  FVar* d_var = nullptr;
  int TX = 1, TY = 16, ccl_tiles = 0;
  int fast_tiles = 0;  // tiles of the fast chain at its smallest tile shape (sparse labels: no LDS bound from nz)
  size_t ccl_lds = 0;
  hipGraphExec_t graph_exec[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // kernel chain with 1 / 2 radix passes, per flag plane
  bool pending = false, search_empty = false;
  bool ref_now = false;  // the search being collected delivers its cells in the reference's order (cfg.reference_order 1, or 2 and small)
  // fast path: _search_end returns as soon as the cluster records have arrived; the kernels that regroup the
  // cells and ship them to the host are still running then.  Everything that reads the cell lists waits here.
  mutable bool cells_fetch = false;  // the grouped cells of the last search are still only on the device
  mutable u32 cells_fetch_n = 0;
  bool lazy_kept = false;  // committed clusters whose host cell lists still sit in the pinned result buffer
  mutable bool tail_pending = false;
  bool fast_launched = false;  // the chain of the running search is the fast one
  u32 epoch = 0;
  int fast_menu = 0;  // menu entry of the running search
  bool fast_ok = false;        // this finder may use the fast chain (decided at creation)
  bool fresh_pending = false;  // fuelmi_frontier_reset not yet executed on the flag plane
  int n_fast = 0, n_legacy = 0, n_fallback = 0;
  int n_in_launch = 0;  // fast-chain searches resolved by the last workgroup of k_tile_cross (the others: by k_resolve)
  int fuse_skip = 0;    // searches left for which k_tile_cross is launched without the in-launch resolve
  bool fuse_tried = false;
  bool kr_skip_ok = false;  // the last search was resolved in the launch with room to spare: k_resolve is not queued for the next one
  bool kr_queued = true;    // ... whether it was for the running search
  u32 rcap_used = 0;        // tile roots the running search's k_tile_cross launch holds
  int menu_min = 0, menu_min_ttl = 0;  // smallest menu entry the next searches start on (after a re-tiled search), and for how long
  int n_retiled = 0;        // searches that ran the chain again on a smaller tile after a per-tile capacity overflow
  int n_late_resolve = 0;   // searches that outgrew that guess (k_resolve + k_tile_out queued by _search_end)
  // cell order of the searches (fuelmi_frontier_order_stats): what the last one delivered (0 address order, 1 the
  // reference's BFS order), how many delivered the reference's, how many wanted it (cfg.reference_order == 2) and fell
  // back to the address order because a cluster exceeded FR_REFORDER_AUTO cells, and that cluster's size
  int order_last = 0, n_order_ref = 0, n_order_fallback = 0;
  u32 order_fallback_cells = 0;
  double wait_us_acc = 0.0;  // host time spent polling for results inside _search_end (fuelmi_bench_host_profile)
  size_t tile_lds[4] = {0, 0, 0, 0}, cross_lds[4] = {0, 0, 0, 0}, out_lds[4] = {0, 0, 0, 0}, fast_items[4] = {0, 0, 0, 0};
  size_t resolve_lds = 0;  // (dynamic LDS of the fast chain's kernels, per tile of the menu)
  std::unique_ptr<StageScope> scope;
  bool dirty_all = true;    // flags / occupancy changed outside the updated-box bookkeeping
  unsigned seen_epoch = 0;  // map->occ_epoch at the last completed search
  int rm_lo[3], rm_hi[3];   // index box of the clusters removed by the current search (rm_lo > rm_hi: none)
  fuelmi_viewpoint_cfg vcfg;
  bool have_vcfg = false;
  // clusters being tested by the running search ("did a cell stop being a frontier cell?"): the verdicts
  // arrive with the search result, the lists are updated in _search_end
  struct PendingRm {
    std::list<HCluster>* list;
    std::list<HCluster>::iterator it;
    int pos;  // position in its list when the search began
  };
  std::vector<PendingRm> pend_rm;
  int* h_changed = nullptr;  // pinned verdicts
  size_t h_changed_cap = 0;
  void* h_cand = nullptr;    // pinned (pool offset, first index) table of the candidates, read by the kernels
  int* d_mark = nullptr;     // device: search number in which candidate k was last found changed
  int rm_mark = 0;
  u32* rm_bar = nullptr;     // device: arrival counter of k_rm_pool_bar's in-kernel barrier (never reset: rm_bar_total is its target)
  u32 rm_bar_total = 0;
  bool rm_failed = false;    // the barrier of k_rm_pool_bar timed out in the search being collected (fuelmi_frontier_search_end fails)
  void* h_put = nullptr;     // pinned table of k_pool_put
  size_t h_put_cap = 0;
  u32* pool = nullptr;  // device copies of the cells of frontiers_ / dormant_frontiers_
  size_t pool_cap = 0, pool_used = 0;
  int last_fin = 1;     // which multisplit buffer holds the grouped cells of the last search
  struct SplitScratch* split = nullptr;  // device buffers of the split stage (frontier_split.hip)
  struct OrderScratch* order = nullptr;  // device buffers of the reference-order stage (frontier_order.hip)
};
void frontier_split_free(fuelmi_frontier* f);
void frontier_order_free(fuelmi_frontier* f);
// cfg.reference_order: the cells of every kept cluster of this search in the order FrontierFinder::expandFrontier
// discovers them (frontier_finder.cpp:123-164), an NQ seed first.  in: the grouped result of the search
// (F.ms_val[fin] / F.ms_key[fin], records F.h_rec).  out: F.ms_val[1 - fin] / F.ms_key[1 - fin] hold *n_total =
// n_out + (clusters started by an NQ seed) cells, cluster r at h_off2[r] .. h_off2[r + 1] (host vector).
int frontier_reference_order(fuelmi_frontier* f, u32 nq, u32 nkept, u32 n_out, int fin, u32* n_total,
                             std::vector<u32>* h_off2, bool fetch_cells);
// tmp cluster -> committed: materialises the host list and copies the cells into the device pool
struct PoolPut {  // one cluster's copy into the cell pool
  u64 dst;
  u32 src, n;
  int seed, pad;
};
int frontier_keep_clusters(fuelmi_frontier* f, std::list<HCluster>& clusters);
int frontier_materialize_lists(fuelmi_frontier* f);
// wait for the tail of the last search (cell regrouping + copy-out); cheap when nothing is pending
static inline int frontier_tail_sync(const fuelmi_frontier* f) {
  if (!f->tail_pending) return FUELMI_OK;
  f->tail_pending = false;
  // poll: the tail is a few microseconds of work that has usually finished long before anybody asks, and a
  // blocking stream synchronisation costs ~15 us of wake-up latency even then
  for (;;) {
    const hipError_t q = hipStreamQuery(f->stream);
    if (q == hipSuccess) return FUELMI_OK;
    if (q != hipErrorNotReady) HIPCHK(q);
  }
}
// ... and make sure the grouped cell lists are in the pinned result buffer (the lazy clusters point into it)
static inline int frontier_prev_ready(const fuelmi_frontier* f) {
  if (!f->prev_pending) return FUELMI_OK;
  for (;;) {
    const hipError_t q = hipEventQuery(f->ev_prev);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) HIPCHK(q);
  }
  const_cast<fuelmi_frontier*>(f)->prev_pending = false;
  return FUELMI_OK;
}
static inline int frontier_cells_ready(const fuelmi_frontier* f) {
  const int rc = frontier_tail_sync(f);
  if (rc) return rc;
  if (f->cells_fetch) {
    f->cells_fetch = false;
    HIPCHK(hipMemcpy(f->F.h_cells, f->F.ms_val[f->last_fin], (size_t)f->cells_fetch_n * sizeof(u32), hipMemcpyDeviceToHost));
  }
  return FUELMI_OK;
}
// between _search_begin and _search_end the cluster lists belong to the running search (its verdicts on
// changed clusters are applied in _search_end): calls that would modify them are refused
#define FRONTIER_NOT_SEARCHING(f, what)                                                        \
  do {                                                                                         \
    if ((f)->pending) {                                                                        \
      fuelmi_set_error(what ": a search is in flight (call fuelmi_frontier_search_end first)"); \
      return FUELMI_EINVAL;                                                                    \
    }                                                                                          \
  } while (0)


// runs the stable radix multisplit of F2.ms_key[0]/ms_val[0] (F2.counts[0] items, F2.counts[3] keys,
// key records F2.krec already laid out), the per-cluster accumulators and the copy-out to the pinned
// host buffers of F2, on f->stream (defined in frontier.hip; used by the split stage with its own
// argument block)
int frontier_regroup(fuelmi_frontier* f, const FArgs& F2, int npass);
// splitLargeFrontiers on the device (frontier_split.hip)
// in: the unsplit result of this search (nkept clusters, n_out grouped Q0 cells in F.ms_val[fin] with
// their cluster rank in F.ms_key[fin], records in F.h_rec).  out: F.h_counts/h_rec/h_part/h_cells hold
// the clusters after splitting (*n_final clusters, *n_cells cells, NQ seeds included as last cell of
// their cluster); filtered[r] = down-sampled cells of final cluster r.
int frontier_split_run(fuelmi_frontier* f, u32 nq, u32 nkept, u32 n_out, int fin, u32* n_final, u32* n_cells,
                       std::vector<std::vector<float>>* filtered);

#endif
