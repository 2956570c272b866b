/*
 * fuelmi.h -- C-ABI of libfuelmi.so: the MI355X (gfx950) implementation of FUEL's per-cycle
 * mapping-and-planning hot path.  Plain C types only (no torch / Eigen / ROS in any signature).
 *
 * The reference has no FFI layer: the path sits behind three C++ class APIs
 * (fast_planner::SDFMap / EDTEnvironment, FrontierFinder, BsplineOptimizer).  The C++ facade in
 * fuel_amd/facade/ re-declares those classes over this C-ABI.  Every entry
 * point below names the reference interface it replaces; paths are relative to
 * /root/reference/fuel_planner/.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FUELMI_E* code on failure, and never
 *     throws or aborts; fuelmi_last_error() returns a thread-local message for the last failure.
 *   - voxel linear address: adr = x*ny*nz + y*nz + z  (plan_env/include/plan_env/sdf_map.h:145-147)
 *   - one HIP stream per map; mutators of one map must be called from one thread at a time
 *     (the reference runs them on the single ros::spin thread); different maps are independent.
 *     Host-staged queries (dist_grad, coarse_dist, query_state, sync_host) may run from other threads
 *     beside a mutator: they and the fusion entry points share the map's staging buffers under a per-map mutex.
 *     Ordering of a distance query (dist_grad, coarse_dist, the one-shot B-spline calls) against ESDF updates of the
 *     same map: it sees the field either before or after an update, never a mix -- a query issued while an update is
 *     queued or running waits for it on the device, an update queued behind launched query kernels waits for them,
 *     and a reader / writer lock keeps a query from slipping between the two (with signed_dist a reader can
 *     therefore never see the positive-only intermediate the negative pass merges into in place).
 *   - all work is done by HIP kernels; there is no CPU fallback.  If no gfx950 device is
 *     usable fuelmi_map_create fails with FUELMI_ENODEV.
 */
#ifndef FUELMI_H_
#define FUELMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FUELMI_OK 0
#define FUELMI_EINVAL (-1)  /* bad argument */
#define FUELMI_ENODEV (-2)  /* no usable HIP device */
#define FUELMI_EHIP (-3)    /* HIP runtime error (message in fuelmi_last_error) */
#define FUELMI_ENOMEM (-4)
#define FUELMI_ELIMIT (-5)  /* problem exceeds a documented limit */

/* Process-level set-up: hardware queues.  A map owns one HIP stream, a finder two, every busy query thread one; the HIP
 * runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share one
 * time-slice (INTEGRATION.md "streams and queues").  The runtime latches the variable at its first call, so the robust
 * way is to export GPU_MAX_HW_QUEUES=16 where the process is launched.  fuelmi_init() is the explicit in-process
 * alternative (rounds 4-5 did this from a load-time constructor; that is now opt-in, FUELMI_SET_HW_QUEUES=1): called
 * BEFORE the process's first HIP call and before threads that touch the environment exist, it puts
 * GPU_MAX_HW_QUEUES=<hw_queues, 16 if <= 0> into the environment unless a value is already there.  Idempotent; always
 * returns FUELMI_OK and reports through fuelmi_hw_queues_state() what happened -- including FUELMI_HWQ_LATE when the HIP
 * runtime was already up (it is detected through the runtime's open /dev/kfd handle; one line on stderr unless
 * FUELMI_QUIET is set): the streams then share the runtime's default of 4 queues.  fuelmi_map_create says so once on
 * stderr when neither the environment nor fuelmi_init() decided. */
#define FUELMI_HWQ_UNINIT 0 /* fuelmi_init() not called: the runtime sees the environment as it is */
#define FUELMI_HWQ_SET 1    /* set by fuelmi_init() before the HIP runtime initialised: in effect */
#define FUELMI_HWQ_ENV 2    /* the environment already held a value: kept */
#define FUELMI_HWQ_LATE 3   /* the HIP runtime was initialised first without the variable: its default (4) is in effect */
int fuelmi_init(int hw_queues);
/* the number of hardware queues the process's HIP runtime uses, as far as the library can know it: the environment's
 * value (4 if unset), or 4 in state FUELMI_HWQ_LATE */
int fuelmi_hw_queues(void);
int fuelmi_hw_queues_state(void);
const char* fuelmi_last_error(void);
const char* fuelmi_version(void);
/* number of visible HIP devices (0 if none / runtime unusable) */
int fuelmi_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Map: replaces fast_planner::SDFMap  (plan_env/include/plan_env/sdf_map.h:27-84)
 * ---------------------------------------------------------------------------------------- */

/* the ROS parameters SDFMap::initMap reads (plan_env/src/sdf_map.cpp:19-47,78-82) */
typedef struct {
  double resolution;          /* sdf_map/resolution */
  double map_size[3];         /* sdf_map/map_size_{x,y,z} */
  double ground_height;       /* sdf_map/ground_height */
  double obstacles_inflation; /* sdf_map/obstacles_inflation */
  double local_bound_inflate; /* sdf_map/local_bound_inflate */
  double default_dist;        /* sdf_map/default_dist */
  int optimistic;             /* sdf_map/optimistic */
  int signed_dist;            /* sdf_map/signed_dist */
  double p_hit, p_miss, p_min, p_max, p_occ;
  double max_ray_length;      /* sdf_map/max_ray_length */
  double virtual_ceil_height; /* sdf_map/virtual_ceil_height */
  double box_min[3], box_max[3]; /* sdf_map/box_{min,max}_{x,y,z} (exploration box) */
  int device;                 /* HIP device ordinal this map lives on */
} fuelmi_map_cfg;

/* derived constants, as initMap computes them (sdf_map.cpp:30-56,78-84) */
typedef struct {
  int voxel_num[3];
  double origin[3];
  double min_boundary[3], max_boundary[3];
  double resolution_inv;
  int box_min[3], box_max[3];           /* posToIndex of the exploration box */
  double prob_hit_log, prob_miss_log, clamp_min_log, clamp_max_log, min_occupancy_log;
  int inflate_step;                     /* ceil(obstacles_inflation / resolution) */
} fuelmi_map_info;

typedef struct fuelmi_map fuelmi_map;

/* SDFMap::initMap (sdf_map.cpp:12-93): allocates the device grid, all voxels unknown */
int fuelmi_map_create(const fuelmi_map_cfg* cfg, fuelmi_map** out);
void fuelmi_map_destroy(fuelmi_map* m);
int fuelmi_map_get_info(const fuelmi_map* m, fuelmi_map_info* info);

/* SDFMap::inputPointCloud (sdf_map.cpp:259-345).  xyz: n points, float x,y,z at the start of
 * each stride_bytes record (12 for packed, 16 for pcl::PointXYZ); host memory. */
int fuelmi_map_input_points(fuelmi_map* m, const float* xyz, int stride_bytes, int n,
                            const double camera_pos[3]);
/* Depth-image front end of the fusion: MapROS::proessDepthImage (plan_env/src/map_ros.cpp:176-215) and
 * the part of MapROS::depthPoseCallback (:121-150) around it.  Parameters are the map_ros/... ROS
 * parameters of the same names (map_ros.cpp:22-30). */
typedef struct {
  double fx, fy, cx, cy;
  double depth_filter_maxdist, depth_filter_mindist;
  int depth_filter_margin;
  double k_depth_scaling_factor; /* raw 16-bit depth units per metre (1000 for millimetres) */
  int skip_pixel;
} fuelmi_depth_cfg;
/* depth: rows x cols row-major 16UC1 image; cam_q_wxyz: camera orientation quaternion (pose->orientation, w
 * first).  The image may lie in pageable host memory (a cv::Mat: staged through a pinned buffer, free again on
 * return), in pinned / registered host memory (fuelmi_host_register: read in place over PCIe) or in device memory
 * (read in place); in the last two cases the caller leaves it unchanged until its next call on this map.  Projects on the device and fuses the points without a host round
 * trip; a frame taken from outside the map is ignored like the reference does.  *n_points (may be
 * NULL) receives proj_points_cnt.  Follow with fuelmi_map_inflate_local (local_updated_ branch). */
int fuelmi_host_register(void* ptr, size_t bytes); /* hipHostRegister(mapped) of a frame ring; undo with _unregister
                                                     * BEFORE the memory is freed (a registration that outlives its
                                                     * memory makes later copies from that address range fail) */
int fuelmi_host_unregister(void* ptr);
/* plain device buffers for callers without a HIP runtime of their own (tests, bench.py) */
int fuelmi_device_alloc(int device, size_t bytes, void** out);
int fuelmi_device_upload(void* dst, const void* src, size_t bytes);
int fuelmi_device_free(void* ptr);
/* STREAM-triad over three device arrays of `bytes` each: the HBM bandwidth a plain kernel reaches on this device
 * (reported by bench.py beside the vendor peak the rooflines are quoted against). */
int fuelmi_hbm_triad(int device, size_t bytes, int reps, double* gb_per_s);
/* the same for the x pass's traffic mix: bytes_in of 16-bit values read, 2 * bytes_in of floats written; reports
 * 3 * bytes_in / time */
int fuelmi_hbm_expand(int device, size_t bytes_in, int reps, double* gb_per_s);
int fuelmi_map_input_depth(fuelmi_map* m, const unsigned short* depth, int rows, int cols,
                           const fuelmi_depth_cfg* cfg, const double cam_pos[3], const double cam_q_wxyz[4],
                           int* n_points);
/* projection only: the reference's point_cloud_[0..proj_points_cnt) as packed float xyz (host). */
int fuelmi_map_project_depth(fuelmi_map* m, const unsigned short* depth, int rows, int cols,
                             const fuelmi_depth_cfg* cfg, const double cam_pos[3], const double cam_q_wxyz[4],
                             float* xyz, int cap, int* n_points);
/* SDFMap::clearAndInflateLocalMap (sdf_map.cpp:434-471) over the current local bound */
int fuelmi_map_inflate_local(fuelmi_map* m);
/* SDFMap::updateESDF3d (sdf_map.cpp:152-241) over the current local bound */
int fuelmi_map_update_esdf(fuelmi_map* m);
/* Which kernel family runs the update (all are exact -- they differ in how far the outward scans of the y / x passes
 * look).  AUTO (default): chosen per update from the far-output statistic of the slabs the local
 * bound covers; PLAIN: the packed 16-bit z/y pass (falls back to PLAIN32 for boxes it does not cover: z extents above
 * 255 voxels, nz % 4 != 0); FAR: the far-field kernels (block / line minima); PLAIN32: the 32-bit plain z/y pass.
 * fuelmi_map_last_esdf_family returns the family the z/y pass of the last update actually ran (tests assert the
 * choice instead of a duration). */
enum { FUELMI_ESDF_AUTO = -1, FUELMI_ESDF_PLAIN = 0, FUELMI_ESDF_FAR = 1, FUELMI_ESDF_PLAIN32 = 2 };
int fuelmi_map_set_esdf_family(fuelmi_map* m, int family);
int fuelmi_map_last_esdf_family(const fuelmi_map* m);
/* which inflation kernels the last fuelmi_map_inflate_local ran: 0 the fused single launch, 1 the factored y/z + x pair
 * (chosen by the size of the box's address range; -1 before the first call).  For tests that must know which code ran. */
int fuelmi_map_last_inflate_kernel(const fuelmi_map* m);
/* SDFMap::resetBuffer() (sdf_map.cpp:95-99) and resetBuffer(min,max) (:101-114) */
int fuelmi_map_reset_buffer_all(fuelmi_map* m);
int fuelmi_map_reset_buffer(fuelmi_map* m, const double min_pos[3], const double max_pos[3]);
/* SDFMap::setOccupied (sdf_map.h:210-215) for n positions; occ must be 0 or 1 */
int fuelmi_map_set_occupied(fuelmi_map* m, const double* pos_xyz, int n, int occ);
/* md_->local_bound_min_/max_ (inclusive voxel indices).  The reference sets them inside
 * inputPointCloud / resetBuffer(); the setter lets a caller run "full-box" updates. */
int fuelmi_map_get_local_bound(const fuelmi_map* m, int bmin[3], int bmax[3]);
int fuelmi_map_set_local_bound(fuelmi_map* m, const int bmin[3], const int bmax[3]);
/* SDFMap::getUpdatedBox (sdf_map.cpp:491-495); setter for callers that fuse elsewhere */
int fuelmi_map_get_updated_box(fuelmi_map* m, double bmin[3], double bmax[3], int reset);
int fuelmi_map_set_updated_box(fuelmi_map* m, const double bmin[3], const double bmax[3]);

/* Bulk load of occupancy_buffer_ (log-odds, double[N], host) -- replaces nothing in the
 * reference (its buffers are host vectors); used to restore a map / build benchmarks. */
int fuelmi_map_upload_occupancy(fuelmi_map* m, const double* occ);

/* Host mirrors for the reference's inline getters (sdf_map.h:196-237 read host vectors
 * directly).  Each non-NULL pointer is a full-size host buffer laid out like the reference's
 * (occupancy_buffer_ double[N], occupancy_buffer_inflate_ char[N], distance_buffer_ double[N]);
 * exactly the voxels inside [bmin,bmax] (inclusive indices; NULL = whole map) are refreshed from
 * the device, with one stream synchronisation per call. */
int fuelmi_map_sync_host(fuelmi_map* m, const int bmin[3], const int bmax[3], double* occupancy,
                         char* inflate, double* distance);
/* Optional, once per mirror buffer: pins the caller's full-size buffers where they lie (any may be NULL)
 * and maps them into the device address space.  fuelmi_map_sync_host calls naming a registered buffer
 * then store the box voxels straight into it -- one kernel, box-limited PCIe traffic, one
 * synchronisation -- instead of staging them.  The buffers must outlive the registration
 * (fuelmi_map_unregister_mirrors / fuelmi_map_destroy end it). */
int fuelmi_map_register_mirrors(fuelmi_map* m, double* occupancy, char* inflate, double* distance);
int fuelmi_map_unregister_mirrors(fuelmi_map* m);

/* SDFMap::getDistWithGrad (sdf_map.cpp:497-536) == EDTEnvironment::evaluateEDTWithGrad
 * (plan_env/src/edt_environment.cpp:78-87) for n host positions; re-entrant w.r.t. queries */
int fuelmi_map_dist_grad(fuelmi_map* m, const double* pos_xyz, int n, double* dist, double* grad_xyz);
/* SDFMap::getDistance(pos) == EDTEnvironment::evaluateCoarseEDT(pos,-1) (edt_environment.cpp:89-97) */
int fuelmi_map_coarse_dist(fuelmi_map* m, const double* pos_xyz, int n, double* dist);
/* SDFMap::getOccupancy / getInflateOccupancy for n voxel indices (-1 outside the map) */
int fuelmi_map_query_state(fuelmi_map* m, const int* idx_xyz, int n, int* occupancy, int* inflate);

int fuelmi_map_synchronize(fuelmi_map* m);

/* ------------------------------------------------------------------------------------------
 * Frontier scan + clustering: replaces the grid part of active_perception::FrontierFinder
 * (active_perception/src/frontier_finder.cpp:54-164 searchFrontiers/expandFrontier,
 *  :353-390 haveOverlap/isFrontierChanged/computeFrontierInfo, :811-881 neighbour helpers)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int cluster_min; /* frontier/cluster_min */
  double min_z;    /* the literal 0.4 at frontier_finder.cpp:151 */
  /* splitLargeFrontiers (frontier_finder.cpp:166-242) + computeFrontierInfo's down-sampling (:374-390,757-774) */
  double cluster_size_xy; /* frontier/cluster_size_xy (2.0) */
  int down_sample;        /* frontier/down_sample (3): VoxelGrid leaf = down_sample * resolution */
  int split;              /* 0: search stops before splitLargeFrontiers; 1: it runs, and every new cluster
                             carries its filtered_cells_ */
  int reference_order;    /* 0 (default): a cluster lists its cells in ascending voxel address and its mean is
                             evaluated order-free (exact integer sums) -- the fast path.  1: the reference's own
                             order -- cells in the BFS order of expandFrontier (frontier_finder.cpp:123-164,
                             neighbour order :848-860), average_ as its sequential f64 sum (:374-390), VoxelGrid
                             centroids accumulated in that order (:757-774) -- so that means, filtered_cells_,
                             split pieces and viewpoints reproduce the reference bit for bit; costs one BFS
                             level sweep per cluster on the device and host-side means (clusters of up to 26624
                             cells are swept inside LDS: +0.15 ... 0.6 ms per search as measured on the streaming
                             workload; larger ones through L2: 24 ms for the 140 k cells of a full 400x400x100
                             box).  2 ("auto"): the reference's order for every search whose clusters all hold at
                             most 26624 cells -- the incremental searches of an exploration run -- and the address
                             order for the giant ones */
} fuelmi_frontier_cfg;

typedef struct fuelmi_frontier fuelmi_frontier;

int fuelmi_frontier_create(fuelmi_map* m, const fuelmi_frontier_cfg* cfg, fuelmi_frontier** out);
void fuelmi_frontier_destroy(fuelmi_frontier* f);
/* forget all clusters and clear frontier_flag_ (== constructing a fresh FrontierFinder,
 * frontier_finder.cpp:23-27) */
int fuelmi_frontier_reset(fuelmi_frontier* f);
/* diagnostics: searches answered by the fast clustering chain [0], by the legacy chain [1], and searches that
 * started on the fast chain and fell back because an input exceeded one of its capacities [2] */
int fuelmi_frontier_stats(const fuelmi_frontier* f, int out3[3]);
/* of the searches the fast chain answered: how many were resolved (cross-tile unions, cluster records, the result the
 * caller polls for) by the last workgroup of the chain's second kernel itself rather than by a launch of their own --
 * searches of up to 1 024 tile-local components; for tests that must know which code ran */
int fuelmi_frontier_resolved_in_launch(const fuelmi_frontier* f);
/* the cell order the searches delivered (cfg.reference_order is a request; mode 2 answers per search): [0] the last
 * search's order -- 0 ascending address, 1 the reference's BFS order --, [1] searches that delivered the reference's
 * order, [2] searches of a mode-2 finder that fell back to the address order because a cluster was too large for the
 * in-LDS level sweep, [3] cells of the largest cluster of the last such search.  A caller that needs the reference's
 * bits checks [0] after fuelmi_frontier_search_end (the facade logs the first fallback). */
int fuelmi_frontier_order_stats(const fuelmi_frontier* f, int out4[4]);
/* waits for everything queued on the finder's stream.  fuelmi_frontier_search_end returns as soon as the cluster
 * records have arrived; the regrouping of the cells and their copy to the host finish behind it (calls that read
 * cell lists wait by themselves) */
int fuelmi_frontier_synchronize(fuelmi_frontier* f);
/* Viewpoint sampling and coverage (frontier_finder.cpp:392-423,662-755,697-719; camera frustum
 * perception_utils.cpp:6-19,49-69,84-93).  Fields are the ROS parameters of the same names. */
typedef struct {
  double candidate_rmin, candidate_rmax; /* frontier/candidate_rmin, candidate_rmax */
  int candidate_rnum;                    /* frontier/candidate_rnum */
  double candidate_dphi;                 /* frontier/candidate_dphi */
  double min_candidate_clearance;        /* frontier/min_candidate_clearance */
  int min_visib_num;                     /* frontier/min_visib_num */
  double min_candidate_dist;             /* frontier/min_candidate_dist (used by the Top/ViewpointsInfo getters) */
  double min_view_finish_fraction;       /* frontier/min_view_finish_fraction */
  double top_angle, left_angle, right_angle, max_dist; /* perception_utils/... */
} fuelmi_viewpoint_cfg;
int fuelmi_frontier_set_viewpoint_cfg(fuelmi_frontier* f, const fuelmi_viewpoint_cfg* cfg);
/* computeFrontiersToVisit: samples viewpoints for every new cluster (needs cfg.split, i.e. filtered
 * cells, and the map's CURRENT inflated occupancy); clusters with at least one viewpoint are appended
 * to frontiers_ with their viewpoints sorted by coverage (best first), the others to
 * dormant_frontiers_.  The new-cluster list is left empty. */
int fuelmi_frontier_compute_to_visit(fuelmi_frontier* f, int* n_active_new, int* n_dormant_new);
int fuelmi_frontier_viewpoint_count(const fuelmi_frontier* f, int which, int k);
/* pos_yaw4: x, y, z, yaw per viewpoint; visib: Viewpoint::visib_num_ */
int fuelmi_frontier_viewpoints(const fuelmi_frontier* f, int which, int k, double* pos_yaw4, int* visib);
/* isFrontierCovered against the map's accumulated updated box (the box is not consumed) */
int fuelmi_frontier_is_covered(fuelmi_frontier* f, int* covered);
/* Frontier::filtered_cells_ of cluster k (VoxelGrid centroids, float xyz, ascending leaf index like PCL);
 * empty unless the cluster was found with cfg.split != 0 */
int fuelmi_frontier_cluster_filtered_size(const fuelmi_frontier* f, int which, int k);
int fuelmi_frontier_cluster_filtered(const fuelmi_frontier* f, int which, int k, float* xyz);
/* searchFrontiers up to (not including) splitLargeFrontiers: consumes the map's updated box
 * (getUpdatedBox(reset=true)), drops changed clusters, scans, clusters.  *n_new = number of new
 * clusters (tmp_frontiers_.size()). */
int fuelmi_frontier_search(fuelmi_frontier* f, int* n_new);
/* The same search split in two: _begin queues the test for changed clusters and the device pipeline on
 * the frontier's own HIP stream without waiting; _end waits, drops the changed clusters from
 * frontiers_ / dormant_frontiers_ (removed_ids_) and assembles the new ones.  Work queued on the map
 * between the two calls (inflation, ESDF, B-spline evaluation) overlaps the scan, which only reads the
 * occupancy state.  The next frame MAY be fused between _begin and _end (a streaming pipeline: fuelmi_bench_stream):
 * the search reads the occupancy planes only in its first kernels and the fusion waits for those on the device.
 * _commit, _reset, _compute_to_visit, _is_covered and a second _begin return FUELMI_EINVAL until _end has been called. */
int fuelmi_frontier_search_begin(fuelmi_frontier* f);
int fuelmi_frontier_search_end(fuelmi_frontier* f, int* n_new);
/* Pipelined delivery for callers that work in cycles (fresh search every cycle): with keep_previous on,
 * fuelmi_frontier_reset does not discard the new clusters of the search it retires -- they become list 3 and stay
 * readable (size / cells / centres / info) until the NEXT reset: their cell lists live in the buffer set the reset
 * retires, which nothing touches for a whole cycle.  Reading list 3 waits only for that search's own tail, so cycle
 * k - 1's cells are copied out while cycle k runs on the device. */
int fuelmi_frontier_keep_previous(fuelmi_frontier* f, int on);
/* move tmp_frontiers_ into frontiers_ (dormant=0) or dormant_frontiers_ (dormant=1) */
int fuelmi_frontier_commit(fuelmi_frontier* f, int dormant);
/* which: 0 tmp_frontiers_, 1 frontiers_, 2 dormant_frontiers_, 3 the new clusters of the search before the last reset
 * (fuelmi_frontier_keep_previous) */
int fuelmi_frontier_count(const fuelmi_frontier* f, int which);
int fuelmi_frontier_cluster_size(const fuelmi_frontier* f, int which, int k);
/* cells of cluster k as linear voxel addresses, ascending (the reference keeps BFS order;
 * the SET is identical) */
int fuelmi_frontier_cluster_cells(const fuelmi_frontier* f, int which, int k, int* adr);
/* the same cells as voxel CENTRES, xyz[3 * size] doubles (the storage of the reference's vector<Vector3d> cells_,
 * frontier_finder.h:27): indexToPos of every cell (sdf_map.h:137-140), same order as _cluster_cells.  Clusters of 32 768
 * cells and more are decoded by the calling thread and three helper threads of the library (created on first use, parked
 * on a condition variable between calls, shared by all finders of the process; FUELMI_HOST_HELPERS=0 keeps everything on
 * the caller) */
int fuelmi_frontier_cluster_centres(const fuelmi_frontier* f, int which, int k, double* xyz);
/* average_[3], box_min_[3], box_max_[3] (computeFrontierInfo) */
int fuelmi_frontier_cluster_info(const fuelmi_frontier* f, int which, int k, double out9[9]);
int fuelmi_frontier_removed_count(const fuelmi_frontier* f);
int fuelmi_frontier_removed_ids(const fuelmi_frontier* f, int* ids);
/* frontier_flag_ expanded to one byte per voxel (char[N], host) -- for tests/debug */
int fuelmi_frontier_get_flags(fuelmi_frontier* f, char* flags);

/* ------------------------------------------------------------------------------------------
 * B-spline cost + gradient: replaces BsplineOptimizer::combineCost and the calc*Cost terms
 * (bspline_opt/src/bspline_optimizer.cpp:255-516, 518-691), batched over C trajectories.
 * ---------------------------------------------------------------------------------------- */
#define FUELMI_COST_SMOOTHNESS (1 << 0)
#define FUELMI_COST_DISTANCE (1 << 1)
#define FUELMI_COST_FEASIBILITY (1 << 2)
#define FUELMI_COST_START (1 << 3)
#define FUELMI_COST_END (1 << 4)
#define FUELMI_COST_GUIDE (1 << 5)
#define FUELMI_COST_WAYPOINTS (1 << 6)
#define FUELMI_COST_VIEWCONS (1 << 7)
#define FUELMI_COST_MINTIME (1 << 8)

/* BsplineOptimizer::setParam (bspline_optimizer.cpp:25-53) */
typedef struct {
  double ld_smooth, ld_dist, ld_feasi, ld_start, ld_end, ld_guide, ld_waypt, ld_view, ld_time;
  double dist0, max_vel, max_acc, wnl, dlmin;
  int bspline_degree;
} fuelmi_bspline_cfg;

/* One batch of C independent combineCost evaluations, all with the same cost_function,
 * dimension and point count (host arrays, row-major, candidate-major):
 *   x          [C][nvar]   nvar = dim*N (+1 trailing knot span if MINTIME)     NLopt layout
 *   pt_dist    [C]         optimize()'s pt_dist_ from the INITIAL control points (:136-140)
 *   knot_span  [C]         used when MINTIME is clear
 *   time_lb    [C] or NULL (treated as -1)
 *   start_state[C][3][3]   pos, vel, acc (START)
 *   end_state  [C][3][3]   END; end_n = end_state_.size() in {1,2,3}
 *   guide_pts  [C][N-2*order][3]   (GUIDE)
 *   waypoints  [C][n_waypt][3], waypt_idx [C][n_waypt]   (WAYPOINTS)
 *   view_pt/view_dir [C][3], view_idx [C]                (VIEWCONS)
 * outputs: cost [C], grad [C][nvar]. */
typedef struct {
  int cost_function;
  int dim;
  int point_num;
  int n_traj;
  const double* x;
  const double* pt_dist;
  const double* knot_span;
  const double* time_lb;
  const double* start_state;
  const double* end_state;
  int end_n;
  const double* guide_pts;
  const double* waypoints;
  const int* waypt_idx;
  int n_waypt;
  const double* view_pt;
  const double* view_dir;
  const int* view_idx;
} fuelmi_bspline_batch;

int fuelmi_bspline_cost_grad(fuelmi_map* m, const fuelmi_bspline_cfg* cfg,
                             const fuelmi_bspline_batch* batch, double* cost, double* grad);

/* BsplineOptimizer::optimize() (bspline_optimizer.cpp:165-253) as ONE call for a batch given in host arrays -- what
 * the reference's callers do one trajectory at a time (plan_manage/src/planner_manager.cpp:296-314).  Like
 * fuelmi_bspline_cost_grad it runs on a query slot of the map (own side stream, inputs and results in the slot's pinned
 * block): no device allocation, re-entrant, concurrent callers do not queue behind each other or behind the map's
 * mutators.  Semantics of the solve: fuelmi_bspline_dev_optimize_timed.  x_out [C][nvar], cost_out [C], evals_out [C]
 * or NULL. */
int fuelmi_bspline_optimize(fuelmi_map* m, const fuelmi_bspline_cfg* cfg, const fuelmi_bspline_batch* batch,
                            int max_eval, double max_time_s, double* x_out, double* cost_out, int* evals_out);

/* Device-resident variant for benchmarking / optimiser loops: upload once, evaluate many times
 * without host copies.  Handles are owned by the map. */
typedef struct fuelmi_bspline_dev fuelmi_bspline_dev;
int fuelmi_bspline_dev_create(fuelmi_map* m, const fuelmi_bspline_cfg* cfg,
                              const fuelmi_bspline_batch* batch, fuelmi_bspline_dev** out);
int fuelmi_bspline_dev_eval(fuelmi_bspline_dev* b);             /* async on the map's stream */
int fuelmi_bspline_dev_download(fuelmi_bspline_dev* b, double* cost, double* grad);
/* the evaluation with its results delivered to one of two pinned host slots by the kernel itself (slot 0 / 1);
 * _collect waits for that launch only and copies cost [C] / grad [C][nvar] out */
int fuelmi_bspline_dev_eval_pinned(fuelmi_bspline_dev* b, int slot);
int fuelmi_bspline_dev_collect(fuelmi_bspline_dev* b, int slot, double* cost, double* grad);
/* BsplineOptimizer::optimize() (bspline_optimizer.cpp:165-253) for every candidate of the batch on the
 * device: start point and bounds as the reference sets them up (control points clamped into the
 * exploration box shrunk by 0.1, +-10 around the start, knot span in [0,5]), at most max_eval objective
 * evaluations (set_maxeval), xtol_rel 1e-5, the best variables seen are returned (best_variable_).  The
 * iteration itself is a box-projected L-BFGS instead of NLopt's (third party): final costs are
 * comparable, iterates are not.  x_out [C][nvar], cost_out [C], evals_out [C] or NULL; synchronous. */
int fuelmi_bspline_dev_optimize(fuelmi_bspline_dev* b, int max_eval, double* x_out, double* cost_out,
                                int* evals_out);
/* ... with the wall-clock cap of the reference's solver as well (opt.set_maxtime(max_iteration_time_[...]),
 * bspline_optimizer.cpp:170-172; 5 ms in exploration_manager/launch/algorithm.xml:190): a candidate's solve stops
 * at the first evaluation boundary past max_time_s seconds of device time and returns the best variables seen so far,
 * like costFunction keeps them (:693-707).  max_time_s <= 0: no cap. */
int fuelmi_bspline_dev_optimize_timed(fuelmi_bspline_dev* b, int max_eval, double max_time_s, double* x_out,
                                      double* cost_out, int* evals_out);
void fuelmi_bspline_dev_destroy(fuelmi_bspline_dev* b);

/* Spline glue around the solve, batched (NonUniformBspline, bspline/src/non_uniform_bspline.cpp).
 *
 * fuelmi_bspline_parameterize = parameterizeToBspline (:178-265) for n_traj sample sets at once: the
 * least-squares control points of a uniform B-spline (degree 3..5, knot span ts) through n_points
 * samples whose start/end velocity and acceleration match `derivs` (the reference solves the
 * (K+4) x (K+degree-1) system with Eigen's ColPivHouseholderQR; results agree to ~1e-12).
 *   ts [C], points [C][K][3], derivs [C][4][3] (start vel, end vel, start acc, end acc)
 *   -> ctrl [C][K+degree-1][3].  ts <= 0 is rejected ("time step error", :181-184).
 * fuelmi_bspline_boundary_states = getBoundaryStates(ks, ke) (:107-122) of n_traj uniform B-splines
 * (setUniformBspline :15-32): position and the first ks / ke derivatives at t = 0 / t = duration.
 *   ctrl [C][n_ctrl][3] -> start [C][ks+1][3], end [C][ke+1][3].
 * Host arrays, synchronous. */
int fuelmi_bspline_parameterize(fuelmi_map* m, int n_traj, int n_points, int degree, const double* ts,
                                const double* points, const double* derivs, double* ctrl);
int fuelmi_bspline_boundary_states(fuelmi_map* m, int n_traj, int n_ctrl, int degree, const double* ts,
                                   const double* ctrl, int ks, int ke, double* start, double* end);
/* The planners' sequence "samples -> parameterizeToBspline -> getBoundaryStates(2, 0) ->
 * setBoundaryStates -> optimize" (plan_manage/src/planner_manager.cpp:161-184, 296-314) without leaving
 * the device: refills the batch `b` (dim 3, point_num = n_points + bspline_degree - 1) with the fitted
 * control points, knot span ts, pt_dist_, start state (pos, vel, acc) and end position; asynchronous on
 * the map's stream, the next _dev_eval / _dev_optimize uses them.  Rows 1..2 of end_state keep what
 * the batch was created with (the planners pass end_n = 1). */
/* Measurement driver: n full-box plan cycles issued back to back from C++ (no interpreter between the calls) --
 * per cycle fuelmi_frontier_reset, fuelmi_map_set_updated_box(ub_min, ub_max), fuelmi_frontier_search_begin,
 * fuelmi_map_inflate_local, fuelmi_map_update_esdf, fuelmi_bspline_dev_eval(batch) (batch may be NULL),
 * fuelmi_frontier_search_end; with `serial` != 0 the search runs after the map chain instead of beside it.
 * Both streams are drained before the clock starts and before it stops.  *n_clusters: clusters of the last
 * search; *seconds: elapsed wall time. */
/* The streaming counterpart: n cycles of fuelmi_map_input_depth(depth[k], pose k), fuelmi_frontier_search_begin,
 * (when the frame fused points) fuelmi_map_inflate_local + fuelmi_map_update_esdf, fuelmi_bspline_dev_eval,
 * fuelmi_frontier_search_end, fuelmi_frontier_commit.  depth[k]: any pointer fuelmi_map_input_depth takes;
 * cam_pos3 / cam_q4: n x 3 / n x 4 doubles; *box_voxels (may be NULL): sum of the local-bound volumes. */
int fuelmi_bench_stream(fuelmi_map* m, fuelmi_frontier* f, fuelmi_bspline_dev* batch, int n, const void* const* depth,
                        int rows, int cols, const fuelmi_depth_cfg* cfg, const double* cam_pos3, const double* cam_q4,
                        int serial, int* n_clusters, double* box_voxels, double* seconds);
int fuelmi_bench_cycles(fuelmi_map* m, fuelmi_frontier* f, fuelmi_bspline_dev* batch, const double ub_min[3],
                        const double ub_max[3], int n, int serial, int* n_clusters, double* seconds);
/* mean HOST microseconds per cycle the last fuelmi_bench_cycles run spent inside each C-ABI call: [0] _frontier_reset +
 * _set_updated_box, [1] _search_begin, [2] _inflate_local, [3] _update_esdf, [4] _bspline_dev_eval, [5] _search_end
 * (of which [6] polling for the device's result). */
int fuelmi_bench_host_profile(const fuelmi_map* m, double out7[7]);
/* fuelmi_bench_cycles with the results delivered to host memory every cycle, as the reference's callers receive
 * them (exploration_manager/src/fast_exploration_manager.cpp:99-114 reads the cell lists of searchFrontiers,
 * plan_manage/src/planner_manager.cpp:296-314 the optimiser's cost / gradient): cells of all new clusters into
 * cells_out (as many clusters as fit cells_cap), cost[C] and grad[C * nvar] of the batch.  seconds3: [0] elapsed,
 * [1] spent in the cell copies, [2] in the cost / gradient download. */
int fuelmi_bench_cycles_delivered(fuelmi_map* m, fuelmi_frontier* f, fuelmi_bspline_dev* batch, const double ub_min[3],
                                  const double ub_max[3], int n, int* cells_out, size_t cells_cap, double* cost,
                                  double* grad, int* n_clusters, double* seconds3);
int fuelmi_bspline_dev_load_samples(fuelmi_bspline_dev* b, int n_points, const double* ts, const double* points,
                                    const double* derivs);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): HIP events recorded on the map's own stream.
 * ---------------------------------------------------------------------------------------- */
enum {
  FUELMI_K_INFLATE = 0,
  FUELMI_K_ESDF_ZY = 1,
  FUELMI_K_ESDF_X = 2,
  FUELMI_K_FRONTIER = 3,
  FUELMI_K_BSPLINE = 4,
  FUELMI_K_INSERT = 5,
  FUELMI_K_COUNT = 6
};
int fuelmi_timer_begin(fuelmi_map* m);
int fuelmi_timer_end(fuelmi_map* m, float* elapsed_ms); /* synchronises the stream */
/* bracket every launch of the stages selected by stage_mask (bit k = FUELMI_K_*) with events */
int fuelmi_profile_enable(fuelmi_map* m, unsigned stage_mask);
/* synchronises, then returns launches and summed device milliseconds of a stage since enable */
int fuelmi_profile_get(fuelmi_map* m, int stage, int* launches, double* total_ms);
/* the same brackets one by one (up to cap values, milliseconds; *n = how many were written): lets the
 * caller take a median, a single disturbed launch otherwise dominates a short sample */
int fuelmi_profile_get_samples(fuelmi_map* m, int stage, double* ms, int cap, int* n);
/* begin / end of every bracket of a stage, milliseconds after the fuelmi_profile_enable call that armed the stage: the
 * device's timeline of a few cycles without a tracer on the host */
int fuelmi_profile_get_timeline(fuelmi_map* m, int stage, double* begin_ms, double* end_ms, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* FUELMI_H_ */
