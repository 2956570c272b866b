/*
 * fuel_oracle.h -- CPU ORACLE for the FUEL mapping-and-planning hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it.  The product
 * (libfuelmi.so, fuel_amd/) never includes, links or calls anything here.
 *
 * It is a dependency-free, single-threaded, double-precision RESTATEMENT of
 * the reference algorithms (file:line citations are relative to
 * /root/reference/fuel_planner/):
 *   plan_env/include/plan_env/sdf_map.h:86-266      MapParam/MapData, index helpers, getters
 *   plan_env/src/sdf_map.cpp:12-93                  initMap (derived constants, buffers)
 *   plan_env/src/sdf_map.cpp:95-114                 resetBuffer
 *   plan_env/src/sdf_map.cpp:116-241                fillESDF / updateESDF3d
 *   plan_env/src/sdf_map.cpp:243-362                setCacheOccupancy / inputPointCloud / closetPointInMap
 *   plan_env/src/sdf_map.cpp:434-471                clearAndInflateLocalMap
 *   plan_env/src/sdf_map.cpp:491-536                getUpdatedBox / getDistWithGrad
 *   plan_env/src/raycast.cpp:6-23,323-407           RayCaster
 *   plan_env/src/edt_environment.cpp:78-97          evaluateEDTWithGrad / evaluateCoarseEDT
 *   active_perception/src/frontier_finder.cpp:54-164,353-390,811-881   frontier scan + clustering
 *   bspline_opt/src/bspline_optimizer.cpp:110-163,255-516,518-691      cost terms + combineCost
 *
 * PARITY PINNING: the reference ships no golden vectors/tests for this path
 * (SURVEY.md section 4), so the restatement is pinned two ways:
 *   (1) against the REAL reference sources compiled with header shims into
 *       oracle/_ref/ (see oracle/ref_build/), when /root/reference is present;
 *       fixtures produced by that build are committed under tests/golden/;
 *   (2) against independent implementations (scipy EDT, scipy label,
 *       finite differences) in tests/.
 */
#ifndef FUEL_ORACLE_H_
#define FUEL_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors the ROS parameters read in sdf_map.cpp:19-47,78-82 */
typedef struct {
  double resolution;
  double map_size[3];
  double ground_height;
  double obstacles_inflation;
  double local_bound_inflate;
  double default_dist;
  int optimistic;
  int signed_dist;
  double p_hit, p_miss, p_min, p_max, p_occ;
  double max_ray_length;
  double virtual_ceil_height;
  double box_min[3], box_max[3]; /* exploration box (sdf_map/box_*) */
} fo_map_cfg;

typedef struct fo_map fo_map;

fo_map* fo_map_create(const fo_map_cfg* cfg);
void fo_map_destroy(fo_map* m);

/* geometry / derived constants (sdf_map.cpp:30-56) */
void fo_map_voxel_num(const fo_map* m, int out[3]);
void fo_map_origin(const fo_map* m, double out[3]);
void fo_map_box_index(const fo_map* m, int bmin[3], int bmax[3]);
/* out[0..4] = prob_hit_log, prob_miss_log, clamp_min_log, clamp_max_log, min_occupancy_log */
void fo_map_logodds(const fo_map* m, double out[5]);

/* raw buffers (length = nx*ny*nz), borrowed pointers valid until destroy */
double* fo_map_occupancy(fo_map* m);
char* fo_map_inflate(fo_map* m);
double* fo_map_distance(fo_map* m);
double* fo_map_distance_neg(fo_map* m);
char* fo_map_flag_rayend(fo_map* m);

/* SDFMap::inputPointCloud; xyz = n packed float triples with given byte stride */
void fo_map_input_points(fo_map* m, const float* xyz, int stride_bytes, int n, const double cam[3]);
/* SDFMap::clearAndInflateLocalMap over [local_bound_min_, local_bound_max_] */
void fo_map_inflate_local(fo_map* m);
/* SDFMap::updateESDF3d over [local_bound_min_, local_bound_max_] */
void fo_map_update_esdf(fo_map* m);
/* SDFMap::resetBuffer() / resetBuffer(min,max) */
void fo_map_reset_buffer_all(fo_map* m);
void fo_map_reset_buffer(fo_map* m, const double min_pos[3], const double max_pos[3]);
/* SDFMap::setOccupied */
void fo_map_set_occupied(fo_map* m, const double pos[3], int occ);
/* local_bound_ accessors (the reference sets them in inputPointCloud/resetBuffer;
   the setter is the test hook for "full-box mode") */
void fo_map_get_local_bound(const fo_map* m, int bmin[3], int bmax[3]);
void fo_map_set_local_bound(fo_map* m, const int bmin[3], const int bmax[3]);
/* SDFMap::getUpdatedBox; setter is a test hook */
void fo_map_get_updated_box(fo_map* m, double bmin[3], double bmax[3], int reset);
void fo_map_set_updated_box(fo_map* m, const double bmin[3], const double bmax[3]);
/* getters */
int fo_map_get_occupancy_idx(const fo_map* m, const int id[3]);
int fo_map_get_occupancy_pos(const fo_map* m, const double pos[3]);
int fo_map_get_inflate_idx(const fo_map* m, const int id[3]);
double fo_map_get_distance_idx(const fo_map* m, const int id[3]);
/* SDFMap::getDistWithGrad == EDTEnvironment::evaluateEDTWithGrad (static env) */
void fo_map_dist_grad(const fo_map* m, const double* pos, int n, double* dist, double* grad);
/* raycaster walk exposed for tests: cells visited between start(end voxel) and end(camera),
   as inputPointCloud consumes them (first nextId discarded). returns count (<= cap). */
int fo_raycast_cells(const fo_map* m, const double start[3], const double end[3], int* idx_xyz, int cap);

/* ---------------- FrontierFinder (scan + clustering) ---------------- */
typedef struct {
  int cluster_min;         /* frontier/cluster_min */
  double min_z;            /* the hard-coded 0.4 in frontier_finder.cpp:151 */
  double cluster_size_xy;  /* frontier/cluster_size_xy (2.0) */
  int down_sample;         /* frontier/down_sample (3): VoxelGrid leaf = down_sample * resolution; <= 0: skip */
  int split;               /* 0: stop before splitLargeFrontiers (the F1-F4 contract); 1: run it */
  int canonical_order;     /* 0: cells reach the VoxelGrid in BFS order (the reference); 1: in ascending voxel
                              address (the order libfuelmi lists cells in), and cluster means are
                              evaluated order-free from exact integer index sums -- changes only the float
                              summation order inside a leaf and the last bits (~1e-13) of the means */
  int flip_principal_dir;  /* test knob: negate splitHorizontally's first principal direction (the SIGN Eigen's
                              EigenSolver would return is not reproduced by the stand-in; flipping it must only
                              permute the pieces) */
} fo_frontier_cfg;
typedef struct fo_frontier fo_frontier;

fo_frontier* fo_frontier_create(fo_map* m, const fo_frontier_cfg* cfg);
void fo_frontier_destroy(fo_frontier* f);
char* fo_frontier_flags(fo_frontier* f);
/* Viewpoint sampling (frontier_finder.cpp:392-423,662-755) + camera FOV (perception_utils.cpp:6-19,49-69,84-93) */
typedef struct {
  double candidate_rmin, candidate_rmax; /* frontier/candidate_rmin|rmax (1.5, 2.5) */
  int candidate_rnum;                    /* frontier/candidate_rnum (3) */
  double candidate_dphi;                 /* frontier/candidate_dphi (15 * 3.1415926 / 180) */
  double min_candidate_clearance;        /* frontier/min_candidate_clearance (0.21) */
  int min_visib_num;                     /* frontier/min_visib_num (15) */
  double min_candidate_dist;             /* frontier/min_candidate_dist (0.75) */
  double min_view_finish_fraction;       /* frontier/min_view_finish_fraction (0.2) */
  double top_angle, left_angle, right_angle, max_dist; /* perception_utils/... */
} fo_viewpoint_cfg;
void fo_frontier_set_viewpoint_cfg(fo_frontier* f, const fo_viewpoint_cfg* c);
/* computeFrontiersToVisit (:392-423): sampleViewpoints for every tmp cluster; clusters with viewpoints go
   to frontiers_ (viewpoints sorted by visib_num, best first), the others to dormant_frontiers_ */
void fo_frontier_compute_to_visit(fo_frontier* f);
int fo_frontier_viewpoint_count(const fo_frontier* f, int which, int k);
void fo_frontier_viewpoints(const fo_frontier* f, int which, int k, double* pos_yaw4, int* visib);
/* isFrontierCovered (:697-719) against the map's current updated box (not reset) */
int fo_frontier_is_covered(fo_frontier* f);
/* down-sampled cells of a cluster (Frontier::filtered_cells_, frontier_finder.cpp:757-774): count and
   xyz triples in the VoxelGrid's output order */
int fo_frontier_cluster_filtered_size(const fo_frontier* f, int which, int k);
void fo_frontier_cluster_filtered(const fo_frontier* f, int which, int k, double* xyz);
/* searchFrontiers up to (not including) splitLargeFrontiers unless cfg.split: removes changed clusters
   (frontiers_ and dormant_), scans, grows clusters into tmp_frontiers_.  Returns number of
   new clusters (tmp_frontiers_.size()). */
int fo_frontier_search(fo_frontier* f);
/* move tmp_frontiers_ to frontiers_ (what computeFrontiersToVisit does minus viewpoint
   sampling; dormant=1 moves them to dormant_frontiers_) */
void fo_frontier_commit(fo_frontier* f, int dormant);
/* which: 0 = tmp_frontiers_, 1 = frontiers_, 2 = dormant_frontiers_ */
int fo_frontier_count(const fo_frontier* f, int which);
int fo_frontier_cluster_size(const fo_frontier* f, int which, int k);
/* cells of cluster k as linear voxel addresses, BFS order (reference order) */
void fo_frontier_cluster_cells(const fo_frontier* f, int which, int k, int* adr);
/* average_, box_min_, box_max_ (computeFrontierInfo) -> out[9] */
void fo_frontier_cluster_info(const fo_frontier* f, int which, int k, double* out9);
int fo_frontier_removed_count(const fo_frontier* f);
void fo_frontier_removed_ids(const fo_frontier* f, int* ids);

/* ---------------- BsplineOptimizer cost / gradient ---------------- */
typedef struct {
  double ld_smooth, ld_dist, ld_feasi, ld_start, ld_end, ld_guide, ld_waypt, ld_view, ld_time;
  double dist0, max_vel, max_acc, wnl, dlmin;
  int bspline_degree;
} fo_bspline_cfg;

/* One combineCost evaluation (bspline_optimizer.cpp:518-691) for one trajectory.
   x: dim*N (+1 if cost_function & MINTIME) variables in NLopt layout.
   pt_dist: the value optimize() derives from the INITIAL control points (:136-140).
   start_state/end_state: 3 vectors each (pos, vel, acc) packed [3][3]; end_n = end_state_.size().
   guide_pts: (N - 2*order) x 3, waypoints: n_waypt x 3 + idx, view: pt[3],dir[3],idx -- may be NULL
   if the corresponding cost bit is clear.  knot_span is used when MINTIME is clear. */
typedef struct {
  int cost_function;
  int dim;
  int point_num;
  double knot_span;
  double pt_dist;
  double time_lb;
  const double* start_state; /* 9 */
  const double* end_state;   /* 9 */
  int end_n;
  const double* guide_pts;
  const double* waypoints;
  const int* waypt_idx;
  int n_waypt;
  const double* view_pt;  /* 3 */
  const double* view_dir; /* 3 */
  int view_idx;
} fo_bspline_problem;

double fo_bspline_pt_dist(const double* ctrl_pts, int n, int dim);
void fo_bspline_cost_grad(const fo_map* m, const fo_bspline_cfg* cfg, const fo_bspline_problem* pb,
                          const double* x, double* cost, double* grad);

/* BsplineOptimizer::optimize() (bspline_optimizer.cpp:165-253): start point / bounds / best-variable
   tracking / evaluation cap as the reference sets them up around NLopt; the iteration itself (NLopt
   LD_LBFGS, third party, absent) is replaced by a box-projected L-BFGS (memory 8, Armijo backtracking,
   xtol_rel 1e-5) -- PARITY UNPINNED against NLopt, the reference for libfuelmi's device optimiser.
   x_io: start variables in, best variables out; returns the best cost; *evals = objective evaluations. */
double fo_bspline_optimize(const fo_map* m, const fo_bspline_cfg* cfg, const fo_bspline_problem* pb, double* x_io,
                           int max_eval, int* evals);

/* NonUniformBspline::parameterizeToBspline (bspline/src/non_uniform_bspline.cpp:178-265): least-squares
   control points [(K+degree-1) x 3] through K samples with start/end velocity and acceleration
   (derivs4: start vel, end vel, start acc, end acc); the solver is a restatement of column-pivoted
   Householder QR (Eigen, third party).  Returns -1 on the inputs the reference refuses. */
int fo_spline_parameterize(double ts, const double* pts, int K, const double* derivs4, int degree, double* ctrl);
/* setUniformBspline (:15-32) + getBoundaryStates(ks, ke) (:107-122): start [(ks+1) x 3], end [(ke+1) x 3] */
void fo_spline_boundary_states(const double* ctrl, int n, int degree, double ts, int ks, int ke, double* start,
                               double* end);

/* MapROS::proessDepthImage (plan_env/src/map_ros.cpp:176-215); parameters map_ros.cpp:22-30 */
typedef struct {
  double fx, fy, cx, cy;
  double depth_filter_maxdist, depth_filter_mindist;
  int depth_filter_margin;
  double k_depth_scaling_factor;
  int skip_pixel;
} fo_depth_cfg;
/* returns proj_points_cnt; at most cap points are written */
int fo_project_depth(const unsigned short* img, int rows, int cols, const fo_depth_cfg* c, const double pos[3],
                     const double quat_wxyz[4], float* xyz, int cap);

#ifdef __cplusplus
}
#endif
#endif
