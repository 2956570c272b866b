// bspline.hip -- batched B-spline cost + gradient (BsplineOptimizer::combineCost,
// bspline_opt/src/bspline_optimizer.cpp:518-691, and the calc*Cost terms :255-516).
//
// One 64-lane wavefront per candidate trajectory; lane i owns control points i, i+64, ...
// Control points are staged in LDS; every gradient is evaluated in GATHER form (each point sums
// the stencils it belongs to) so no atomics are needed; cost and the knot-span gradient are
// wave-reduced with DPP shuffles.  All arithmetic is f64 like the reference; the ESDF is read
// as 8 f32 gathers per control point (trilinear, SDFMap::getDistWithGrad sdf_map.cpp:497-536).
#include <cmath>
#include <cstring>
#include <vector>

#include "fuelmi_internal.h"

struct BsplineArgs {
  fuelmi_bspline_cfg cfg;
  int cost_function, dim, N, C, end_n, n_waypt, nvar, order, n_guide;
  const double* x;
  const double* pt_dist;
  const double* knot_span;
  const double* time_lb;
  const double* start_state;
  const double* end_state;
  const double* guide_pts;
  const double* waypoints;
  const int* waypt_idx;
  const double* view_pt;
  const double* view_dir;
  const int* view_idx;
  double* cost;
  double* grad;
};

struct fuelmi_bspline_dev {
  fuelmi_map* map;  // cleared if the map is destroyed first (then only _destroy is legal)
  int device = 0;
  BsplineArgs a;
  std::vector<void*> allocs;
  size_t lds;       // evaluation scratch of one wave (the solves build on it)
  size_t lds_eval4; // ... plus the partial gradients / costs of the four-wave cost kernel
  double *opt_x = nullptr, *opt_cost = nullptr;  // fuelmi_bspline_dev_optimize outputs
  int* opt_evals = nullptr;
  // fuelmi_bspline_dev_eval_pinned: two pinned result slots (cost [C] | grad [C][nvar]) the cost kernel writes
  // directly, and the event behind each launch
  double* pin_out[2] = {nullptr, nullptr};
  hipEvent_t ev_out[2] = {nullptr, nullptr};
  double* fit_in = nullptr;  // fuelmi_bspline_dev_load_samples staging: ts | points | derivs
  size_t fit_cap = 0;
};

// 64-lane sum on the DPP data path (no LDS crossbar round trips): quads, half rows, rows, then the two
// row broadcasts of GFX9; lane 63 ends with the total.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_take<0x140, 0xF>(v);  // row_mirror: every lane of a row holds the row's sum
  v += dpp_take<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_take<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double dot3(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// combineCost for candidate c at the variables x (NLopt layout); writes grad[nvar], returns the
// cost in every lane.  One wavefront; smem_raw = 5*3*N doubles of LDS.
// per-candidate constants of the objective, staged once per kernel in LDS behind the 5 x [N][3] scratch
// (a solve evaluates ~200 times: re-reading them from global memory was a dependent load per use)
#define EVAL_CONST 24  // [0] pt_dist [1] knot_span [2] time_lb [3..11] start state [12..20] end state
__device__ __forceinline__ double* eval_const(const BsplineArgs& A, unsigned char* smem_raw) {
  return reinterpret_cast<double*>(smem_raw) + 15 * (size_t)A.N;
}
__device__ void load_eval_const(const BsplineArgs& A, int c, unsigned char* smem_raw) {
  double* K = eval_const(A, smem_raw);
  const int l = threadIdx.x;
  if (l == 0) K[0] = A.pt_dist[c];
  if (l == 1) K[1] = A.knot_span ? A.knot_span[c] : 0.0;
  if (l == 2) K[2] = A.time_lb ? A.time_lb[c] : -1.0;
  if (l >= 3 && l < 12) K[l] = ((A.cost_function & FUELMI_COST_START) && A.start_state) ? A.start_state[(size_t)c * 9 + (l - 3)] : 0.0;
  if (l >= 12 && l < 21) K[l] = ((A.cost_function & FUELMI_COST_END) && A.end_state) ? A.end_state[(size_t)c * 9 + (l - 12)] : 0.0;
  __syncthreads();
}
// NWV = waves that share one candidate: 1 (the solves: a wave per candidate walks its whole objective) or 4 (the batched
// cost kernel: the terms of the objective are dealt to four waves -- smoothness / feasibility / distance / boundary
// and the rest -- which evaluate side by side and add their partial costs and gradients in a fixed order; the cost
// kernel's duration is the latency of one wave's dependent chain, and the chain is a quarter as long).
// NWV = 4 needs 12 N + 8 more doubles of LDS behind the constants.
template <int NWV>
__device__ double bspline_eval(const Geo& g, const float* __restrict__ dist, const BsplineArgs& A, int c,
                               const double* x, double* grad, unsigned char* smem_raw) {
  const double* K = eval_const(A, smem_raw);
  const int N = A.N, dim = A.dim;
  const int wv = NWV == 1 ? 0 : (int)(threadIdx.x >> 6);
  const bool is_sm = NWV == 1 || wv == 0, is_fe = NWV == 1 || wv == 1, is_di = NWV == 1 || wv == 2, is_mi = NWV == 1 || wv == 3;
  double* q = reinterpret_cast<double*>(smem_raw);  // [N][3]
  double* tj = q + 3 * N;                            // [N][3] 2*jerk/pt_dist      (j <= N-4)
  double* tv = tj + 3 * N;                           // [N][3] vel hinge factor    (j <= N-2)
  double* ta = tv + 3 * N;                           // [N][3] acc hinge factor    (j <= N-3)
  double* gw = ta + 3 * N;                           // [N][3] waypoint gradient scratch
  const int lane = threadIdx.x & 63;
  const bool opt_time = (A.cost_function & FUELMI_COST_MINTIME) != 0;
  const double dt = opt_time ? x[A.nvar - 1] : K[1];
  const double pt_dist = K[0];
  const fuelmi_bspline_cfg& P = A.cfg;

  for (int i = threadIdx.x; i < N; i += 64 * NWV)
    for (int j = 0; j < 3; ++j) q[3 * i + j] = (j < dim) ? x[dim * i + j] : 0.0;
  __syncthreads();

  double cost = 0.0, gt = 0.0;  // lane-partial weighted cost and knot-span gradient
  const double dt_inv = 1 / dt, dt_inv2 = dt_inv * dt_inv;
  // Divisions by per-evaluation constants are multiplications by their reciprocals (one f64 division is a
  // dependent chain of ~11 instructions and a solve is latency-bound): each affected term moves by <= 1 ulp
  // against the reference's quotient, 10 orders of magnitude inside the parity bar.
  const double pt_inv = 1 / pt_dist;
  const double r_2dt = 1 / (2 * dt), r_dt2 = 1 / (dt * dt), r_ndt2 = 1 / (-dt * dt), r_ndt3 = 1 / (-dt * dt * dt);
  // the ESDF corners of this lane's first control point: loads start here and are consumed in pass 2,
  // behind the arithmetic of pass 1
  DistGather G0;
  const bool gather0 = is_di && (A.cost_function & FUELMI_COST_DISTANCE) && lane < N;
  if (gather0) dist_gather_issue(g, dist, &q[3 * lane], G0);

  // ---- pass 1: per-stencil quantities ----
  for (int i = lane; i < N; i += 64) {
    for (int k = 0; k < 3; ++k) {
      if (is_sm) tj[3 * i + k] = 0.0;
      if (is_fe) tv[3 * i + k] = 0.0;
      if (is_fe) ta[3 * i + k] = 0.0;
      if (is_mi) gw[3 * i + k] = 0.0;
    }
    if (is_sm && (A.cost_function & FUELMI_COST_SMOOTHNESS) && i + 3 < N) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) {
        double ji = (q[3 * (i + 3) + k] - 3 * q[3 * (i + 2) + k] + 3 * q[3 * (i + 1) + k] - q[3 * i + k]) * pt_inv;
        s += ji * ji;
        tj[3 * i + k] = 2 * ji * pt_inv;
      }
      cost += P.ld_smooth * s;
    }
    if (is_fe && (A.cost_function & FUELMI_COST_FEASIBILITY)) {
      if (i + 1 < N) {
        for (int k = 0; k < 3; ++k) {
          double vi = (q[3 * (i + 1) + k] - q[3 * i + k]) * dt_inv;
          double vd = fabs(vi) - P.max_vel;
          if (vd > 0.0) {
            cost += P.ld_feasi * (vd * vd);
            double sign = vi > 0 ? 1.0 : -1.0;
            double tmp = 2 * vd * sign * dt_inv;
            tv[3 * i + k] = tmp;
            if (opt_time) gt += P.ld_feasi * (tmp * (-vi));
          }
        }
      }
      if (i + 2 < N) {
        for (int k = 0; k < 3; ++k) {
          double ai = (q[3 * (i + 2) + k] - 2 * q[3 * (i + 1) + k] + q[3 * i + k]) * dt_inv2;
          double ad = fabs(ai) - P.max_acc;
          if (ad > 0.0) {
            cost += P.ld_feasi * (ad * ad);
            double sign = ai > 0 ? 1.0 : -1.0;
            double tmp = 2 * ad * sign * dt_inv2;
            ta[3 * i + k] = tmp;
            if (opt_time) gt += P.ld_feasi * (tmp * ai * (-2) * dt);
          }
        }
      }
    }
  }
  __syncthreads();
  if (is_mi && (A.cost_function & FUELMI_COST_WAYPOINTS) && lane == 0) {
    double s = 0.0;
    for (int w = 0; w < A.n_waypt; ++w) {
      const double* wp = A.waypoints + ((size_t)c * A.n_waypt + w) * 3;
      int idx = A.waypt_idx[(size_t)c * A.n_waypt + w];
      for (int k = 0; k < 3; ++k) {
        double dq = 1 / 6.0 * (q[3 * idx + k] + 4 * q[3 * (idx + 1) + k] + q[3 * (idx + 2) + k]) - wp[k];
        s += dq * dq;
        gw[3 * idx + k] += dq * (2.0 / 6.0);
        gw[3 * (idx + 1) + k] += dq * (8.0 / 6.0);
        gw[3 * (idx + 2) + k] += dq * (2.0 / 6.0);
      }
    }
    cost += P.ld_waypt * s;
  }
  __syncthreads();

  // ---- pass 2: per-point gradient (gather) ----
  for (int i = lane; i < N; i += 64) {
    double gq[3] = {0.0, 0.0, 0.0};
    if (is_sm && (A.cost_function & FUELMI_COST_SMOOTHNESS)) {
      // point i sits at offset s of stencil j = i - s; weights (-1, 3, -3, 1)
      const double wgt[4] = {-1.0, 3.0, -3.0, 1.0};
      for (int s = 0; s < 4; ++s) {
        int j = i - s;
        if (j >= 0 && j + 3 < N)
          for (int k = 0; k < 3; ++k) gq[k] += P.ld_smooth * wgt[s] * tj[3 * j + k];
      }
    }
    if (is_di && (A.cost_function & FUELMI_COST_DISTANCE)) {
      double dg[3];
      double d = (i == lane && gather0) ? dist_gather_finish(g, G0, dg) : dist_with_grad_dev(g, dist, &q[3 * i], dg);
      double nrm = sqrt(dot3(dg, dg));
      if (nrm > 1e-4) {
        const double nrm_inv = 1 / nrm;
        for (int k = 0; k < 3; ++k) dg[k] *= nrm_inv;
      }
      if (d < P.dist0) {
        cost += P.ld_dist * ((d - P.dist0) * (d - P.dist0));
        for (int k = 0; k < 3; ++k) gq[k] += P.ld_dist * (2.0 * (d - P.dist0) * dg[k]);
      }
    }
    if (is_fe && (A.cost_function & FUELMI_COST_FEASIBILITY)) {
      for (int k = 0; k < 3; ++k) {
        double s = 0.0;
        if (i + 1 < N) s += -tv[3 * i + k];
        if (i >= 1) s += tv[3 * (i - 1) + k];
        if (i + 2 < N) s += ta[3 * i + k];
        if (i >= 1 && i + 1 < N) s += -2 * ta[3 * (i - 1) + k];
        if (i >= 2) s += ta[3 * (i - 2) + k];
        gq[k] += P.ld_feasi * s;
      }
    }
    // calcStartCost (:355-391) and calcEndCost (:393-431) share their algebra: position (1,4,1)/6,
    // velocity (-1,0,1)/2dt, acceleration (1,-2,1)/dt^2 of three consecutive points against a target
    // state.  The three START lanes and the three END lanes run it together (one pass through the ~12
    // f64 divisions instead of two); the summation order of each term is the reference's ("first" is
    // q1 for START and q_1 for END).  A lane holding both roles (N < 6) takes a second pass.
    {
      const bool is_start = is_mi && (A.cost_function & FUELMI_COST_START) && i < 3;
      const bool is_end = is_mi && (A.cost_function & FUELMI_COST_END) && i >= N - 3;
      for (int pass = 0; pass < 2; ++pass) {
        const bool as_end = (pass == 0) ? (!is_start && is_end) : (is_start && is_end);
        if (!(pass == 0 ? (is_start || is_end) : as_end)) continue;
        const int b0 = as_end ? N - 3 : 0, r = i - b0;  // r: position of this lane's point among the three
        const double* tgt = K + (as_end ? 12 : 3);
        const double w_pos = as_end ? 1.0 : 10.0, lam = as_end ? P.ld_end : P.ld_start;
        const bool have_vel = !as_end || A.end_n >= 2, have_acc = !as_end || A.end_n == 3;
        double c_b = 0.0, gt_b = 0.0;
        for (int k = 0; k < 3; ++k) {
          const double qa = q[3 * b0 + k], qb = q[3 * (b0 + 1) + k], qc = q[3 * (b0 + 2) + k];
          const double first = as_end ? qc : qa, last = as_end ? qa : qc;
          double dq = 1 / 6.0 * (first + 4 * qb + last) - tgt[k];
          c_b += w_pos * dq * dq;
          double gk = w_pos * 2 * dq * ((r == 1) ? (4 / 6.0) : (1 / 6.0));
          if (have_vel) {
            dq = r_2dt * (qc - qa) - tgt[3 + k];
            c_b += dq * dq;
            if (r == 0) gk += 2 * dq * (-1.0) * r_2dt;
            if (r == 2) gk += 2 * dq * 1.0 * r_2dt;
            gt_b += dq * (qc - qa) * r_ndt2;
          }
          if (have_acc) {
            dq = r_dt2 * (first - 2 * qb + last) - tgt[6 + k];
            c_b += dq * dq;
            gk += 2 * dq * ((r == 1) ? -2.0 : 1.0) * r_dt2;
            gt_b += dq * (first - 2 * qb + last) * r_ndt3;
          }
          gq[k] += lam * gk;
        }
        if (r == (as_end ? 2 : 0)) {
          cost += lam * c_b;
          if (opt_time) gt += lam * gt_b;
        }
      }
    }
    if (is_mi && (A.cost_function & FUELMI_COST_GUIDE) && i >= A.order && i < N - A.order) {
      const double* gp = A.guide_pts + ((size_t)c * A.n_guide + (i - A.order)) * 3;
      for (int k = 0; k < 3; ++k) {
        double d = q[3 * i + k] - gp[k];
        cost += P.ld_guide * d * d;
        gq[k] += P.ld_guide * 2 * d;
      }
    }
    if (is_mi && (A.cost_function & FUELMI_COST_WAYPOINTS))
      for (int k = 0; k < 3; ++k) gq[k] += P.ld_waypt * gw[3 * i + k];
    if (is_mi && (A.cost_function & FUELMI_COST_VIEWCONS) && i == A.view_idx[c]) {
      const double* p = A.view_pt + (size_t)c * 3;
      const double* dir = A.view_dir + (size_t)c * 3;
      double dn_ = sqrt(dot3(dir, dir));
      double v[3] = {dir[0] / dn_, dir[1] / dn_, dir[2] / dn_};
      double qp[3] = {q[3 * i] - p[0], q[3 * i + 1] - p[1], q[3 * i + 2] - p[2]};
      double pr = dot3(qp, v);
      double dn[3], dl[3];
      for (int k = 0; k < 3; ++k) {
        dn[k] = qp[k] - pr * v[k];
        dl[k] = pr * v[k];
      }
      double c_view = dot3(dn, dn);
      double vdn = dot3(v, dn);
      for (int k = 0; k < 3; ++k) gq[k] += P.ld_view * 2 * (dn[k] - vdn * v[k]);
      double norm_dl = sqrt(dot3(dl, dl));
      if (norm_dl < dn_) {
        c_view += P.wnl * (norm_dl - dn_) * (norm_dl - dn_);
        double vdl = dot3(v, dl);
        for (int k = 0; k < 3; ++k) gq[k] += P.ld_view * (P.wnl * 2 * (norm_dl - dn_) * (vdl * v[k]) / norm_dl);
      }
      cost += P.ld_view * c_view;
    }
    if (NWV == 1) {
      for (int j = 0; j < dim; ++j) grad[dim * i + j] = gq[j];
    } else {  // this wave's share of the point's gradient
      double* gpart = reinterpret_cast<double*>(smem_raw) + 15 * (size_t)N + EVAL_CONST + (size_t)wv * 3 * N;
      for (int j = 0; j < 3; ++j) gpart[3 * i + j] = gq[j];
    }
  }

  if (is_mi && (A.cost_function & FUELMI_COST_MINTIME) && lane == 0) {
    // calcTimeCost (:504-516)
    double duration = (N - A.order) * dt;
    double cst = duration, g_t = double(N - A.order);
    double lb = K[2];
    if (lb > 0 && duration < lb) {
      const double w_lb = 10;
      cst += w_lb * (duration - lb) * (duration - lb);
      g_t += w_lb * 2 * (duration - lb) * (N - A.order);
    }
    cost += P.ld_time * cst;
    gt += P.ld_time * g_t;
  }
  cost = wave_sum(cost);
  gt = wave_sum(gt);
  if (NWV == 1) {
    if (lane == 0 && opt_time) grad[A.nvar - 1] = gt;
    __syncthreads();  // grad[] complete and visible to every lane
    return cost;
  }
  // the four waves' shares, added in wave order (smoothness, feasibility, distance, the rest)
  double* part = reinterpret_cast<double*>(smem_raw) + 15 * (size_t)N + EVAL_CONST;
  double* cpart = part + 12 * (size_t)N;
  if (lane == 0) cpart[wv] = cost, cpart[4 + wv] = gt;
  __syncthreads();
  for (int e = threadIdx.x; e < dim * N; e += 64 * NWV) {
    const int i = e / dim, j = e - i * dim;
    grad[e] = ((part[3 * i + j] + part[3 * N + 3 * i + j]) + part[6 * N + 3 * i + j]) + part[9 * N + 3 * i + j];
  }
  if (threadIdx.x == 0 && opt_time) grad[A.nvar - 1] = ((cpart[4] + cpart[5]) + cpart[6]) + cpart[7];
  const double total = ((cpart[0] + cpart[1]) + cpart[2]) + cpart[3];
  __syncthreads();
  return total;
}

// four waves per candidate (see bspline_eval): LDS = the evaluation scratch + 12 N + 8 doubles
__global__ void __launch_bounds__(256)
k_bspline_cost_grad(Geo g, const float* __restrict__ dist, BsplineArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int c = blockIdx.x;
  load_eval_const(A, c, smem_raw);
  const double cost = bspline_eval<4>(g, dist, A, c, A.x + (size_t)c * A.nvar, A.grad + (size_t)c * A.nvar, smem_raw);
  if (threadIdx.x == 0) A.cost[c] = cost;
}

// ---------------------------------------------------------------------------------------------
// BsplineOptimizer::optimize() (bspline_optimizer.cpp:165-253) for a whole batch on the device: one
// wavefront per candidate runs the complete solve -- variable clamping and bounds exactly as the
// reference sets them up (:177-214: box shrunk by 0.1, +-10 around the start, knot span in [0,5]),
// best-so-far tracking like costFunction (:693-707), evaluation cap like set_maxeval, xtol_rel 1e-5.
// The reference hands the iteration to NLopt (LD_LBFGS / LD_TNEWTON, third party); here it is a
// box-projected L-BFGS (memory 8, Armijo backtracking) -- the same algorithm as the oracle's
// fo_bspline_optimize, so results agree with it up to reduction order; against NLopt only the final
// cost is comparable.  All work vectors (6 + 2*8 of n ~ 100 doubles) live in LDS behind the
// evaluation scratch: the solve is a chain of ~200 dependent evaluations and dot products, so its
// speed is the latency of every step, not bandwidth.
// ---------------------------------------------------------------------------------------------
struct LbfgsArgs {
  int max_eval;
  unsigned long long max_ticks;  // wall-clock budget of the solve in 100 MHz ticks (0: none) -- NLopt's set_maxtime
  double box_lo[3], box_hi[3];  // exploration box shrunk by 0.1 (:174-178)
  double* x_out;                // [C][nvar] best variables
  double* cost_out;             // [C]
  int* evals_out;               // [C]
};
#define LBFGS_MEM 8

__device__ __forceinline__ double wdot(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += a[i] * b[i];
  return wave_sum(s);
}

__global__ void __launch_bounds__(64)
k_bspline_optimize(Geo g, const float* __restrict__ dist, BsplineArgs A, LbfgsArgs L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int c = blockIdx.x, lane = threadIdx.x, n = A.nvar;
  double* w = reinterpret_cast<double*>(smem_raw) + 15 * (size_t)A.N + EVAL_CONST;  // behind bspline_eval's scratch
  load_eval_const(A, blockIdx.x, smem_raw);
  double *q = w, *gq = q + n, *xn = gq + n, *gn = xn + n, *d = gn + n, *best = d + n;
  double* S = best + n;             // [MEM][n]
  double* Y = S + LBFGS_MEM * n;    // [MEM][n]
  double* rho = Y + LBFGS_MEM * n;  // [MEM]
  double* al = rho + LBFGS_MEM;     // [MEM]
  const double* x0 = A.x + (size_t)c * n;
  const int npt = A.dim * A.N;
  auto lb_of = [&](int i, double q0) {
    if (A.dim == 1) return -1e300;
    if (i >= npt) return 0.0;
    return fmax(q0 - 10.0, L.box_lo[i % 3]);
  };
  auto ub_of = [&](int i, double q0) {
    if (A.dim == 1) return 1e300;
    if (i >= npt) return 5.0;
    return fmin(q0 + 10.0, L.box_hi[i % 3]);
  };
  // start point: control points clamped into the shrunk box (:194-199); d[] keeps it for the bounds
  for (int i = lane; i < n; i += 64) {
    double v = x0[i];
    if (A.dim != 1 && i < npt) v = fmax(fmin(v, L.box_hi[i % 3]), L.box_lo[i % 3]);
    q[i] = v;
    best[i] = v;
  }
  __syncthreads();
  // the bounds refer to the (clamped) START values: recomputed from x0 where needed
  auto start_val = [&](int i) {
    double v = x0[i];
    if (A.dim != 1 && i < npt) v = fmax(fmin(v, L.box_hi[i % 3]), L.box_lo[i % 3]);
    return v;
  };
  int evals = 0;
  // the time cap is checked like NLopt does, between evaluations; the best variables seen so far are what comes back
  // (costFunction's best_variable_, bspline_optimizer.cpp:693-707).  wall_clock64 is the 100 MHz constant clock.
  const unsigned long long t_begin = wall_clock64();
  auto out_of_time = [&]() { return L.max_ticks != 0ull && wall_clock64() - t_begin > L.max_ticks; };
  double f = bspline_eval<1>(g, dist, A, c, q, gq, smem_raw);
  ++evals;
  double fbest = f;
  int hist = 0, head = 0;  // number of stored pairs, slot of the oldest
  while (evals < L.max_eval && !out_of_time()) {
    // projected steepest-descent seed, then the two-loop recursion
    for (int i = lane; i < n; i += 64) {
      const double sv = start_val(i), lo = lb_of(i, sv), hi = ub_of(i, sv);
      const bool at_lb = q[i] <= lo && gq[i] > 0, at_ub = q[i] >= hi && gq[i] < 0;
      d[i] = (at_lb || at_ub) ? 0.0 : -gq[i];
    }
    __syncthreads();
    for (int k = hist - 1; k >= 0; --k) {
      const int sl = (head + k) % LBFGS_MEM;
      const double a = wdot(S + sl * n, d, n) * rho[sl];
      if (lane == 0) al[sl] = a;
      for (int i = lane; i < n; i += 64) d[i] -= a * Y[sl * n + i];
      __syncthreads();
    }
    if (hist > 0) {
      const int sl = (head + hist - 1) % LBFGS_MEM;
      const double yy = wdot(Y + sl * n, Y + sl * n, n), sy = 1.0 / rho[sl];
      const double gamma = yy > 0 ? sy / yy : 1.0;
      for (int i = lane; i < n; i += 64) d[i] *= gamma;
      __syncthreads();
    }
    for (int k = 0; k < hist; ++k) {
      const int sl = (head + k) % LBFGS_MEM;
      const double b = wdot(Y + sl * n, d, n) * rho[sl];
      const double a = al[sl];
      for (int i = lane; i < n; i += 64) d[i] += S[sl * n + i] * (a - b);
      __syncthreads();
    }
    double gd = wdot(gq, d, n);
    if (!(gd < 0)) {  // not a descent direction: restart with steepest descent
      hist = 0, head = 0;
      for (int i = lane; i < n; i += 64) d[i] = -gq[i];
      __syncthreads();
      gd = -wdot(gq, gq, n);
      if (gd == 0) break;
    }
    double step = hist == 0 ? 1.0 / fmax(1.0, sqrt(-gd)) : 1.0, fn = f;
    bool ok = false;
    for (int ls = 0; ls < 20 && evals < L.max_eval && !out_of_time(); ++ls) {
      for (int i = lane; i < n; i += 64) {
        const double sv = start_val(i);
        xn[i] = fmin(fmax(q[i] + step * d[i], lb_of(i, sv)), ub_of(i, sv));
      }
      __syncthreads();
      fn = bspline_eval<1>(g, dist, A, c, xn, gn, smem_raw);
      ++evals;
      if (fn < fbest) {  // costFunction's best_variable_ (:699-703)
        fbest = fn;
        for (int i = lane; i < n; i += 64) best[i] = xn[i];
      }
      double dec = 0.0;
      for (int i = lane; i < n; i += 64) dec += gq[i] * (xn[i] - q[i]);
      dec = wave_sum(dec);
      if (fn <= f + 1e-4 * dec) {
        ok = true;
        break;
      }
      step *= 0.5;
    }
    if (!ok) break;
    // curvature pair into the ring buffer
    double sy = 0.0, ss = 0.0, xx = 0.0;
    const int slot = hist < LBFGS_MEM ? (head + hist) % LBFGS_MEM : head;
    for (int i = lane; i < n; i += 64) {
      const double si = xn[i] - q[i], yi = gn[i] - gq[i];
      sy += si * yi, ss += si * si, xx += xn[i] * xn[i];
    }
    sy = wave_sum(sy), ss = wave_sum(ss), xx = wave_sum(xx);
    if (sy > 1e-12) {
      for (int i = lane; i < n; i += 64) {
        S[slot * n + i] = xn[i] - q[i];
        Y[slot * n + i] = gn[i] - gq[i];
      }
      if (lane == 0) rho[slot] = 1.0 / sy;
      if (hist < LBFGS_MEM)
        ++hist;
      else
        head = (head + 1) % LBFGS_MEM;
    }
    for (int i = lane; i < n; i += 64) {
      q[i] = xn[i];
      gq[i] = gn[i];
    }
    f = fn;
    __syncthreads();
    if (sqrt(ss) <= 1e-5 * sqrt(xx)) break;  // xtol_rel 1e-5
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) L.x_out[(size_t)c * n + i] = best[i];
  if (lane == 0) {
    L.cost_out[c] = fbest;
    L.evals_out[c] = evals;
  }
}

template <int NPL>
__device__ __forceinline__ double reg_dot(const double (&a)[NPL], const double (&b)[NPL]) {
  double s2 = 0.0;
#pragma unroll
  for (int e = 0; e < NPL; ++e) s2 += a[e] * b[e];
  return wave_sum(s2);
}
// objective at register-resident variables: staged through LDS (xs in, gs out) around bspline_eval
template <int NPL, int NWV>
__device__ __forceinline__ double reg_objective(const Geo& g, const float* __restrict__ dist, const BsplineArgs& A, int c,
                                                double* xs, double* gs, unsigned char* smem_raw, const bool (&on)[NPL],
                                                const double (&xv)[NPL], double (&gv)[NPL]) {
  const int lane = threadIdx.x & 63;
  if (NWV == 1 || threadIdx.x < 64) {  // (every wave holds the same variables: one writes them)
#pragma unroll
    for (int e = 0; e < NPL; ++e)
      if (on[e]) xs[lane + 64 * e] = xv[e];
  }
  __syncthreads();
  const double fv = bspline_eval<NWV>(g, dist, A, c, xs, gs, smem_raw);  // ends with a barrier
#pragma unroll
  for (int e = 0; e < NPL; ++e) gv[e] = on[e] ? gs[lane + 64 * e] : 0.0;
  return fv;
}

// The same solve with the whole L-BFGS state in registers: lane l owns the variables l, l+64, ...
// (NPL per lane, n <= 64*NPL), the 2 x 8 history vectors included (ordered oldest -> newest, shifted
// when full, so every index is static).  The workgroup is ONE wavefront: dot products are a multiply-add
// per owned variable plus a DPP reduction, updates are register arithmetic, nothing but the objective
// touches LDS (variables in, gradient out).  Same operations in the same order as the LDS version above
// (which stays as the path for n > 256): identical results, ~1/3 of the time per iteration.
// NWV = 4 (round 4): FOUR wavefronts per candidate.  Every wave carries the same L-BFGS state and does the same
// register arithmetic (identical values, so the control flow is uniform across the workgroup without any exchange);
// what is shared out is the objective -- bspline_eval<4>: smoothness / feasibility / distance / the rest, a wave each
// -- which is where a solve spends its time (a chain of ~100-200 dependent evaluations).
template <int NPL, int NWV>
__global__ void __launch_bounds__(64 * NWV)
k_bspline_optimize_r(Geo g, const float* __restrict__ dist, BsplineArgs A, LbfgsArgs L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_oot;
  const int c = blockIdx.x, lane = threadIdx.x & 63, n = A.nvar;
  // variables handed to the objective: behind its scratch (and behind the partial sums of the four-wave evaluation)
  double* xs = reinterpret_cast<double*>(smem_raw) + 15 * (size_t)A.N + EVAL_CONST + (NWV == 1 ? 0 : 12 * (size_t)A.N + 8);
  load_eval_const(A, blockIdx.x, smem_raw);
  double* gs = xs + n;                                                  // its gradient
  const double* x0 = A.x + (size_t)c * n;
  const int npt = A.dim * A.N;
  bool on[NPL];
  double lo[NPL], hi[NPL], q[NPL], gq[NPL], xn[NPL], gn[NPL], d[NPL], best[NPL];
  double S[LBFGS_MEM][NPL], Y[LBFGS_MEM][NPL], rho[LBFGS_MEM], al[LBFGS_MEM];
#pragma unroll
  for (int e = 0; e < NPL; ++e) {
    const int i = lane + 64 * e;
    on[e] = i < n;
    double v = on[e] ? x0[i] : 0.0;
    // start point: control points clamped into the shrunk box (:194-199); the bounds refer to it
    if (on[e] && A.dim != 1 && i < npt) v = fmax(fmin(v, L.box_hi[i % 3]), L.box_lo[i % 3]);
    q[e] = best[e] = v;
    lo[e] = hi[e] = 0.0;
    if (on[e]) {
      if (A.dim == 1) {
        lo[e] = -1e300, hi[e] = 1e300;
      } else if (i >= npt) {
        lo[e] = 0.0, hi[e] = 5.0;
      } else {
        lo[e] = fmax(v - 10.0, L.box_lo[i % 3]), hi[e] = fmin(v + 10.0, L.box_hi[i % 3]);
      }
    }
    gq[e] = xn[e] = gn[e] = d[e] = 0.0;
#pragma unroll
    for (int k = 0; k < LBFGS_MEM; ++k) S[k][e] = Y[k][e] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < LBFGS_MEM; ++k) rho[k] = al[k] = 0.0;
  int evals = 0;
  // the time cap is checked like NLopt does, between evaluations; the best variables seen so far are what comes back
  // (costFunction's best_variable_, bspline_optimizer.cpp:693-707).  wall_clock64 is the 100 MHz constant clock.
  const unsigned long long t_begin = wall_clock64();
  auto out_of_time = [&]() {
    if (L.max_ticks == 0ull) return false;
    if (NWV == 1) return wall_clock64() - t_begin > L.max_ticks;
    if (threadIdx.x == 0) s_oot = wall_clock64() - t_begin > L.max_ticks ? 1 : 0;  // (one clock for the whole workgroup)
    __syncthreads();
    const bool o = s_oot != 0;
    __syncthreads();
    return o;
  };
  double f = reg_objective<NPL, NWV>(g, dist, A, c, xs, gs, smem_raw, on, q, gq);
  ++evals;
  double fbest = f;
  int hist = 0;  // stored pairs, slot 0 the oldest
  while (evals < L.max_eval && !out_of_time()) {
#pragma unroll
    for (int e = 0; e < NPL; ++e) {
      const bool at_lb = q[e] <= lo[e] && gq[e] > 0, at_ub = q[e] >= hi[e] && gq[e] < 0;
      d[e] = (at_lb || at_ub) ? 0.0 : -gq[e];
    }
#pragma unroll
    for (int k = LBFGS_MEM - 1; k >= 0; --k)
      if (k < hist) {
        const double a = reg_dot<NPL>(S[k], d) * rho[k];
        al[k] = a;
#pragma unroll
        for (int e = 0; e < NPL; ++e) d[e] -= a * Y[k][e];
      }
#pragma unroll
    for (int k = 0; k < LBFGS_MEM; ++k)
      if (k == hist - 1) {
        const double yy = reg_dot<NPL>(Y[k], Y[k]), sy = 1.0 / rho[k];
        const double gamma = yy > 0 ? sy / yy : 1.0;
#pragma unroll
        for (int e = 0; e < NPL; ++e) d[e] *= gamma;
      }
#pragma unroll
    for (int k = 0; k < LBFGS_MEM; ++k)
      if (k < hist) {
        const double b = reg_dot<NPL>(Y[k], d) * rho[k];
        const double a = al[k];
#pragma unroll
        for (int e = 0; e < NPL; ++e) d[e] += S[k][e] * (a - b);
      }
    double gd = reg_dot<NPL>(gq, d);
    if (!(gd < 0)) {  // not a descent direction: restart with steepest descent
      hist = 0;
#pragma unroll
      for (int e = 0; e < NPL; ++e) d[e] = -gq[e];
      gd = -reg_dot<NPL>(gq, gq);
      if (gd == 0) break;
    }
    double step = hist == 0 ? 1.0 / fmax(1.0, sqrt(-gd)) : 1.0, fn = f;
    bool ok = false;
    for (int ls = 0; ls < 20 && evals < L.max_eval && !out_of_time(); ++ls) {
#pragma unroll
      for (int e = 0; e < NPL; ++e) xn[e] = on[e] ? fmin(fmax(q[e] + step * d[e], lo[e]), hi[e]) : 0.0;
      fn = reg_objective<NPL, NWV>(g, dist, A, c, xs, gs, smem_raw, on, xn, gn);
      ++evals;
      if (fn < fbest) {  // costFunction's best_variable_ (:699-703)
        fbest = fn;
#pragma unroll
        for (int e = 0; e < NPL; ++e) best[e] = xn[e];
      }
      double dec = 0.0;
#pragma unroll
      for (int e = 0; e < NPL; ++e) dec += gq[e] * (xn[e] - q[e]);
      dec = wave_sum(dec);
      if (fn <= f + 1e-4 * dec) {
        ok = true;
        break;
      }
      step *= 0.5;
    }
    if (!ok) break;
    double sy = 0.0, ss = 0.0, xx = 0.0, sn[NPL], yn[NPL];
#pragma unroll
    for (int e = 0; e < NPL; ++e) {
      sn[e] = xn[e] - q[e], yn[e] = gn[e] - gq[e];
      sy += sn[e] * yn[e], ss += sn[e] * sn[e], xx += xn[e] * xn[e];
    }
    sy = wave_sum(sy), ss = wave_sum(ss), xx = wave_sum(xx);
    if (sy > 1e-12) {
      if (hist == LBFGS_MEM) {  // drop the oldest pair
#pragma unroll
        for (int k = 0; k + 1 < LBFGS_MEM; ++k) {
          rho[k] = rho[k + 1];
#pragma unroll
          for (int e = 0; e < NPL; ++e) S[k][e] = S[k + 1][e], Y[k][e] = Y[k + 1][e];
        }
        hist = LBFGS_MEM - 1;
      }
#pragma unroll
      for (int k = 0; k < LBFGS_MEM; ++k)
        if (k == hist) {
          rho[k] = 1.0 / sy;
#pragma unroll
          for (int e = 0; e < NPL; ++e) S[k][e] = sn[e], Y[k][e] = yn[e];
        }
      ++hist;
    }
#pragma unroll
    for (int e = 0; e < NPL; ++e) q[e] = xn[e], gq[e] = gn[e];
    f = fn;
    if (sqrt(ss) <= 1e-5 * sqrt(xx)) break;  // xtol_rel 1e-5
  }
  if (NWV > 1 && threadIdx.x >= 64) return;  // (every wave holds the result: one writes it)
#pragma unroll
  for (int e = 0; e < NPL; ++e)
    if (on[e]) L.x_out[(size_t)c * n + lane + 64 * e] = best[e];
  if (lane == 0) {
    L.cost_out[c] = fbest;
    L.evals_out[c] = evals;
  }
}

// ---------------------------------------------------------------------------------------------
// Spline glue around the solve (NonUniformBspline, bspline/src/non_uniform_bspline.cpp): the planners
// turn path samples into control points (parameterizeToBspline :178-265), read the boundary states
// back off the spline (getBoundaryStates :107-122) and hand both to optimize().  One wavefront per
// candidate does all of it out of LDS, so a batch of candidates goes samples -> solve without a host
// round trip.
//
// parameterizeToBspline solves the (K+4) x (K+degree-1) system "spline passes through the samples,
// start/end velocity and acceleration match" in the least-squares sense (Eigen's ColPivHouseholderQR).
// Every row touches `degree` consecutive unknowns, so the normal matrix is banded (half-bandwidth
// degree-1): banded Cholesky plus one step of iterative refinement against the original rows, which
// brings the solution back to QR accuracy (cond(A) < 1e3).
struct FitArgs {
  int C, K, degree;
  const double* ts;      // [C]
  const double* points;  // [C][K][3]
  const double* derivs;  // [C][4][3]  start vel, end vel, start acc, end acc
  double* ctrl;          // candidate c: ctrl + c * stride, (K + degree - 1) rows of 3
  long stride;
  // planner glue (all optional): what setBoundaryStates / optimize() derive from the fitted spline
  int write_dt;          // ctrl[c * stride + 3 n] = ts[c]   (trailing knot-span variable)
  double* knot_span;     // [C]
  double* pt_dist;       // [C]      optimize() :136-140
  double* start_state;   // [C][3][3]  getBoundaryStates(2, 0).start
  double* end_state;     // [C][3][3]  row 0 = getBoundaryStates(2, 0).end[0]
};

// value at t = 0 (at_end = false) or t = duration of the d-th derivative of the uniform B-spline with
// control points q[n][3], degree p, knots u[n + p + 1]  (evaluateDeBoorT of computeDerivatives()[d-1];
// the derivative's knot vector is the parent's without its first and last knot, :99-103)
__device__ void spline_boundary_value(const double* q, int n, int p, const double* u, int d, bool at_end,
                                      double* out) {
  const int pd = p - d, nd = n - d;
  const double duration = u[n] - u[p];  // getTimeSum() of the position spline
  const double uu = (at_end ? duration : 0.0) + u[pd + d];
  const double ub = fmin(fmax(u[pd + d], uu), u[nd + d]);
  int k = pd;
  while (u[k + 1 + d] < ub) ++k;
  const int i0 = k - pd;
  double w[6][3];
  for (int i = 0; i <= p; ++i)
    for (int a = 0; a < 3; ++a) w[i][a] = q[3 * (i0 + i) + a];
  for (int l = 1; l <= d; ++l) {  // getDerivativeControlPoints of level l-1, window only
    const int pl = p - (l - 1);
    for (int i = 0; i + l <= p; ++i) {
      const int gi = i0 + i;
      const double den = u[gi + pl + 1 + (l - 1)] - u[gi + 1 + (l - 1)];
      for (int a = 0; a < 3; ++a) w[i][a] = pl * (w[i + 1][a] - w[i][a]) / den;
    }
  }
  for (int r = 1; r <= pd; ++r)  // de Boor (:51-71), knots of the derivative = u[. + d]
    for (int i = pd; i >= r; --i) {
      const double alpha = (ub - u[i + k - pd + d]) / (u[i + 1 + k - r + d] - u[i + k - pd + d]);
      for (int a = 0; a < 3; ++a) w[i][a] = (1 - alpha) * w[i - 1][a] + alpha * w[i][a];
    }
  for (int a = 0; a < 3; ++a) out[a] = w[pd][a];
}

// setUniformBspline's knot vector (:15-32): cumulative sums from -p * ts
__device__ void spline_uniform_knots(double* u, int n, int p, double ts) {
  const int m = n + p;
  for (int i = 0; i <= m; ++i) u[i] = (i <= p) ? double(-p + i) * ts : u[i - 1] + ts;
}

struct FitRow {
  int c0;           // first unknown the row touches
  const double* w;  // `degree` coefficients
};
__device__ __forceinline__ FitRow fit_row(int r, int K, int degree, const double* wts) {
  if (r < K) return {r, wts};
  const int s = r - K;  // 0 start vel, 1 end vel, 2 start acc, 3 end acc
  return {(s & 1) ? K - 1 : 0, wts + ((s < 2) ? 5 : 10)};
}

// banded Cholesky solve of M x = g for three right-hand sides (lanes 0..2); M holds L afterwards
__device__ void fit_solve(const double* L, int n, int hb, double* g, int lane) {
  if (lane < 3) {
    for (int j = 0; j < n; ++j) {
      double s = g[3 * j + lane];
      for (int d = 1; d <= hb && d <= j; ++d) s -= L[5 * j + d] * g[3 * (j - d) + lane];
      g[3 * j + lane] = s / L[5 * j];
    }
    for (int j = n - 1; j >= 0; --j) {
      double s = g[3 * j + lane];
      for (int d = 1; d <= hb && j + d < n; ++d) s -= L[5 * (j + d) + d] * g[3 * (j + d) + lane];
      g[3 * j + lane] = s / L[5 * j];
    }
  }
}

__global__ __launch_bounds__(64) void k_bspline_fit(FitArgs F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int c = blockIdx.x, lane = threadIdx.x;
  const int K = F.K, p = F.degree, n = K + p - 1, hb = p - 1, R = K + 4;
  double* wts = reinterpret_cast<double*>(smem_raw);  // pos / vel / acc coefficients, 5 each
  double* M = wts + 16;                               // [n][5]  M(j, j-d) at [j][d]
  double* g = M + 5 * n;                              // [n][3]
  double* x = g + 3 * n;                              // [n][3]
  double* res = x + 3 * n;                            // [R][3]
  double* u = res + 3 * R;                            // [n + p + 1]
  const double ts = F.ts[c];
  const double* pts = F.points + (size_t)c * K * 3;
  const double* der = F.derivs + (size_t)c * 12;
  if (lane == 0) {  // coefficient rows as the reference writes them (:199-236)
    for (int i = 0; i < 15; ++i) wts[i] = 0.0;
    if (p == 3) {
      const double a = 1 / 6.0, b = 1 / (2 * ts), e = 1 / (ts * ts);
      wts[0] = a * 1, wts[1] = a * 4, wts[2] = a * 1;
      wts[5] = b * -1, wts[6] = b * 0, wts[7] = b * 1;
      wts[10] = e * 1, wts[11] = e * -2, wts[12] = e * 1;
    } else if (p == 4) {
      const double a = 1 / 24.0, b = 1 / (6 * ts), e = 1 / (2 * ts * ts);
      wts[0] = a * 1, wts[1] = a * 11, wts[2] = a * 11, wts[3] = a * 1;
      wts[5] = b * -1, wts[6] = b * -3, wts[7] = b * 3, wts[8] = b * 1;
      wts[10] = e * 1, wts[11] = e * -1, wts[12] = e * -1, wts[13] = e * 1;
    } else {
      const double cp[5] = {1, 26, 66, 26, 1}, cv[5] = {-1, -10, 0, 10, 1}, ca[5] = {1, 2, -6, 2, 1};
      for (int i = 0; i < 5; ++i) wts[i] = cp[i] / 120.0, wts[5 + i] = cv[i] / (24 * ts), wts[10 + i] = ca[i] / (6 * ts * ts);
    }
  }
  __syncthreads();
  auto rhs = [&](int r, int a) { return r < K ? pts[3 * r + a] : der[3 * (r - K) + a]; };
  // normal matrix and right-hand side, one unknown per lane
  for (int j = lane; j < n; j += 64) {
    double m[5] = {0, 0, 0, 0, 0}, gj[3] = {0, 0, 0};
    auto add = [&](int r) {
      const FitRow fr = fit_row(r, K, p, wts);
      const int o = j - fr.c0;
      if (o < 0 || o >= p) return;
      const double wj = fr.w[o];
      for (int d = 0; d <= hb && d <= o; ++d) m[d] += wj * fr.w[o - d];
      for (int a = 0; a < 3; ++a) gj[a] += wj * rhs(r, a);
    };
    for (int r = max(0, j - p + 1); r <= min(K - 1, j); ++r) add(r);
    for (int r = K; r < R; ++r) add(r);
    for (int d = 0; d < 5; ++d) M[5 * j + d] = m[d];
    for (int a = 0; a < 3; ++a) g[3 * j + a] = gj[a];
  }
  __syncthreads();
  if (lane == 0) {  // banded Cholesky, in place
    for (int j = 0; j < n; ++j) {
      double s = M[5 * j];
      for (int d = 1; d <= hb && d <= j; ++d) s -= M[5 * j + d] * M[5 * j + d];
      const double ljj = sqrt(s);
      M[5 * j] = ljj;
      for (int i = j + 1; i <= j + hb && i < n; ++i) {
        double t = M[5 * i + (i - j)];
        for (int k = max(0, i - hb); k < j; ++k) t -= M[5 * i + (i - k)] * M[5 * j + (j - k)];
        M[5 * i + (i - j)] = t / ljj;
      }
    }
  }
  __syncthreads();
  fit_solve(M, n, hb, g, lane);
  __syncthreads();
  for (int i = lane; i < 3 * n; i += 64) x[i] = g[i];
  __syncthreads();
  // one refinement step: residual of the original rows, normal right-hand side, solve, add
  for (int r = lane; r < R; r += 64) {
    const FitRow fr = fit_row(r, K, p, wts);
    for (int a = 0; a < 3; ++a) {
      double s = rhs(r, a);
      for (int k = 0; k < p; ++k) s -= fr.w[k] * x[3 * (fr.c0 + k) + a];
      res[3 * r + a] = s;
    }
  }
  __syncthreads();
  for (int j = lane; j < n; j += 64) {
    double gj[3] = {0, 0, 0};
    auto add = [&](int r) {
      const FitRow fr = fit_row(r, K, p, wts);
      const int o = j - fr.c0;
      if (o < 0 || o >= p) return;
      for (int a = 0; a < 3; ++a) gj[a] += fr.w[o] * res[3 * r + a];
    };
    for (int r = max(0, j - p + 1); r <= min(K - 1, j); ++r) add(r);
    for (int r = K; r < R; ++r) add(r);
    for (int a = 0; a < 3; ++a) g[3 * j + a] = gj[a];
  }
  __syncthreads();
  fit_solve(M, n, hb, g, lane);
  __syncthreads();
  for (int i = lane; i < 3 * n; i += 64) x[i] += g[i];
  __syncthreads();
  double* out = F.ctrl + (size_t)c * F.stride;
  for (int i = lane; i < 3 * n; i += 64) out[i] = x[i];
  if (F.write_dt && lane == 0) out[3 * n] = ts;
  if (F.knot_span && lane == 0) F.knot_span[c] = ts;
  if (!F.start_state && !F.end_state && !F.pt_dist) return;
  if (lane == 0) spline_uniform_knots(u, n, p, ts);
  __syncthreads();
  if (lane < 3 && F.start_state) spline_boundary_value(x, n, p, u, lane, false, F.start_state + (size_t)c * 9 + 3 * lane);
  if (lane == 3 && F.end_state) spline_boundary_value(x, n, p, u, 0, true, F.end_state + (size_t)c * 9);
  if (lane == 4 && F.pt_dist) {
    double s = 0.0;
    for (int i = 0; i + 1 < n; ++i) {
      const double dx = x[3 * i + 3] - x[3 * i], dy = x[3 * i + 4] - x[3 * i + 1], dz = x[3 * i + 5] - x[3 * i + 2];
      s += sqrt(dx * dx + dy * dy + dz * dz);
    }
    F.pt_dist[c] = s / double(n);
  }
}
static size_t fit_lds(int K, int degree) {
  const int n = K + degree - 1;
  return (size_t)(16 + 5 * n + 3 * n + 3 * n + 3 * (K + 4) + n + degree + 1) * sizeof(double);
}

// getBoundaryStates(ks, ke) of C uniform B-splines: lanes 0..ks evaluate the start states, the next
// ke+1 lanes the end states
__global__ __launch_bounds__(64) void k_bspline_boundary(int n, int p, const double* __restrict__ ts,
                                                         const double* __restrict__ ctrl, int ks, int ke,
                                                         double* __restrict__ start, double* __restrict__ end) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* q = reinterpret_cast<double*>(smem_raw);
  double* u = q + 3 * n;
  const int c = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < 3 * n; i += 64) q[i] = ctrl[(size_t)c * 3 * n + i];
  if (lane == 0) spline_uniform_knots(u, n, p, ts[c]);
  __syncthreads();
  if (lane <= ks) spline_boundary_value(q, n, p, u, lane, false, start + ((size_t)c * (ks + 1) + lane) * 3);
  else if (lane <= ks + 1 + ke)
    spline_boundary_value(q, n, p, u, lane - ks - 1, true, end + ((size_t)c * (ke + 1) + (lane - ks - 1)) * 3);
}

// ---------------------------------------------------------------------------------------------
static int upload(fuelmi_bspline_dev* b, const void* src, size_t bytes, const void** dst) {
  *dst = nullptr;
  if (!src || bytes == 0) return FUELMI_OK;
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, bytes));
  b->allocs.push_back(d);
  HIPCHK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, b->map->stream));
  *dst = d;
  return FUELMI_OK;
}

static void bspline_dev_orphan(void* p) { static_cast<fuelmi_bspline_dev*>(p)->map = nullptr; }  // (its work is on
                                                                                            // the map's stream,
                                                                                            // drained by the map)
extern "C" void fuelmi_bspline_dev_destroy(fuelmi_bspline_dev* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->map) {
    (void)hipStreamSynchronize(b->map->stream);
    std::lock_guard<std::mutex> lk(b->map->dep_mu);
    auto& deps = b->map->dependents;
    for (size_t k = 0; k < deps.size(); ++k)
      if (deps[k].obj == b) {
        deps.erase(deps.begin() + (long)k);
        break;
      }
  }
  for (void* p : b->allocs) (void)hipFree(p);
  for (int k = 0; k < 2; ++k) {
    if (b->pin_out[k]) (void)hipHostFree(b->pin_out[k]);
    if (b->ev_out[k]) (void)hipEventDestroy(b->ev_out[k]);
  }
  delete b;
}

// checks a batch description and derives the sizes combineCost / optimize() work with (pointers stay null)
static int bspline_args_init(const fuelmi_bspline_cfg* cfg, const fuelmi_bspline_batch* in, BsplineArgs& A) {
  ARGCHK(in->dim >= 1 && in->dim <= 3 && in->point_num >= 4 && in->n_traj >= 1);
  ARGCHK(in->x && in->pt_dist);
  const int cf = in->cost_function;
  const bool opt_time = (cf & FUELMI_COST_MINTIME) != 0;
  ARGCHK(opt_time || in->knot_span);
  ARGCHK(!(cf & FUELMI_COST_START) || in->start_state);
  ARGCHK(!(cf & FUELMI_COST_END) || (in->end_state && in->end_n >= 1 && in->end_n <= 3));
  ARGCHK(!(cf & FUELMI_COST_GUIDE) || in->guide_pts);
  ARGCHK(!(cf & FUELMI_COST_WAYPOINTS) || (in->n_waypt >= 0 && (in->n_waypt == 0 || (in->waypoints && in->waypt_idx))));
  ARGCHK(!(cf & FUELMI_COST_VIEWCONS) || (in->view_pt && in->view_dir && in->view_idx));
  memset(&A, 0, sizeof(A));
  A.cfg = *cfg;
  A.cost_function = cf;
  A.dim = in->dim;
  A.N = in->point_num;
  A.C = in->n_traj;
  A.end_n = in->end_n;
  A.n_waypt = (cf & FUELMI_COST_WAYPOINTS) ? in->n_waypt : 0;
  A.nvar = opt_time ? A.dim * A.N + 1 : A.dim * A.N;
  A.order = (A.dim == 1) ? 3 : cfg->bspline_degree;  // optimize() :123-127
  A.n_guide = A.N - 2 * A.order;
  return FUELMI_OK;
}

extern "C" int fuelmi_bspline_dev_create(fuelmi_map* m, const fuelmi_bspline_cfg* cfg,
                                         const fuelmi_bspline_batch* in, fuelmi_bspline_dev** out) {
  ARGCHK(m && cfg && in && out);
  *out = nullptr;
  BsplineArgs A0;
  {
    const int rca = bspline_args_init(cfg, in, A0);
    if (rca) return rca;
  }
  const int cf = in->cost_function;
  const bool opt_time = (cf & FUELMI_COST_MINTIME) != 0;
  HIPCHK(hipSetDevice(m->device));
  fuelmi_bspline_dev* b = new fuelmi_bspline_dev;
  b->map = m;
  b->device = m->device;
  BsplineArgs& A = b->a;
  A = A0;
  (void)opt_time;
  b->lds = ((size_t)A.N * 3 * 5 + EVAL_CONST) * sizeof(double);
  b->lds_eval4 = b->lds + ((size_t)A.N * 12 + 8) * sizeof(double);
  if (b->lds_eval4 > 160 * 1024) {
    fuelmi_set_error("%d control points exceed the LDS budget", A.N);
    delete b;
    return FUELMI_ELIMIT;
  }
  const size_t C = (size_t)A.C;
  int rc = FUELMI_OK;
  auto up = [&](const void* src, size_t bytes, const void** dst) {
    if (rc == FUELMI_OK) rc = upload(b, src, bytes, dst);
  };
  up(in->x, C * A.nvar * sizeof(double), (const void**)&A.x);
  up(in->pt_dist, C * sizeof(double), (const void**)&A.pt_dist);
  up(in->knot_span, C * sizeof(double), (const void**)&A.knot_span);
  up(in->time_lb, C * sizeof(double), (const void**)&A.time_lb);
  if (cf & FUELMI_COST_START) up(in->start_state, C * 9 * sizeof(double), (const void**)&A.start_state);
  if (cf & FUELMI_COST_END) up(in->end_state, C * 9 * sizeof(double), (const void**)&A.end_state);
  if ((cf & FUELMI_COST_GUIDE) && A.n_guide > 0)
    up(in->guide_pts, C * A.n_guide * 3 * sizeof(double), (const void**)&A.guide_pts);
  if (A.n_waypt > 0) {
    up(in->waypoints, C * A.n_waypt * 3 * sizeof(double), (const void**)&A.waypoints);
    up(in->waypt_idx, C * A.n_waypt * sizeof(int), (const void**)&A.waypt_idx);
  }
  if (cf & FUELMI_COST_VIEWCONS) {
    up(in->view_pt, C * 3 * sizeof(double), (const void**)&A.view_pt);
    up(in->view_dir, C * 3 * sizeof(double), (const void**)&A.view_dir);
    up(in->view_idx, C * sizeof(int), (const void**)&A.view_idx);
  }
  if (rc == FUELMI_OK) {
    void* d = nullptr;
    if (hipMalloc(&d, C * sizeof(double)) == hipSuccess) {
      b->allocs.push_back(d);
      A.cost = (double*)d;
    } else
      rc = FUELMI_ENOMEM;
    if (rc == FUELMI_OK && hipMalloc(&d, C * A.nvar * sizeof(double)) == hipSuccess) {
      b->allocs.push_back(d);
      A.grad = (double*)d;
    } else
      rc = FUELMI_ENOMEM;
  }
  if (rc != FUELMI_OK) {
    fuelmi_bspline_dev_destroy(b);
    return rc;
  }
  if (b->lds_eval4 > 64 * 1024) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bspline_cost_grad),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  {
    std::lock_guard<std::mutex> lk(m->dep_mu);
    m->dependents.push_back({b, &bspline_dev_orphan});
  }
  *out = b;
  return FUELMI_OK;
}

extern "C" int fuelmi_bspline_dev_eval(fuelmi_bspline_dev* b) {
  ARGCHK(b);
  HIPCHK(hipSetDevice(b->map->device));
  StageScope sc(b->map, FUELMI_K_BSPLINE);
  k_bspline_cost_grad<<<b->a.C, 256, b->lds_eval4, b->map->stream>>>(b->map->g, b->map->dist, b->a);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

// The evaluation with its results DELIVERED: the kernel writes cost and gradient straight into one of two pinned host
// slots (posted PCIe writes of ~50 KB, no copy engine, no blocking synchronisation); _collect waits for that launch
// only and copies the slot out.  A caller that works in cycles evaluates into slot k & 1 and collects cycle k - 1's
// slot while cycle k runs.
extern "C" int fuelmi_bspline_dev_eval_pinned(fuelmi_bspline_dev* b, int slot) {
  ARGCHK(b && (slot == 0 || slot == 1));
  HIPCHK(hipSetDevice(b->map->device));
  const size_t C = (size_t)b->a.C, n = (size_t)b->a.nvar;
  if (!b->pin_out[slot]) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&b->pin_out[slot]), (C + C * n) * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&b->ev_out[slot], hipEventDisableTiming));
  }
  BsplineArgs A = b->a;
  A.cost = b->pin_out[slot];
  A.grad = b->pin_out[slot] + C;
  {
    StageScope sc(b->map, FUELMI_K_BSPLINE);
    k_bspline_cost_grad<<<A.C, 256, b->lds_eval4, b->map->stream>>>(b->map->g, b->map->dist, A);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(b->ev_out[slot], b->map->stream));
  return FUELMI_OK;
}
extern "C" int fuelmi_bspline_dev_collect(fuelmi_bspline_dev* b, int slot, double* cost, double* grad) {
  ARGCHK(b && (slot == 0 || slot == 1) && cost && grad && b->pin_out[slot]);
  HIPCHK(hipSetDevice(b->map->device));
  for (;;) {
    const hipError_t q = hipEventQuery(b->ev_out[slot]);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) HIPCHK(q);
  }
  const size_t C = (size_t)b->a.C, n = (size_t)b->a.nvar;
  memcpy(cost, b->pin_out[slot], C * sizeof(double));
  memcpy(grad, b->pin_out[slot] + C, C * n * sizeof(double));
  return FUELMI_OK;
}

// whole solves on the device; synchronous: returns the best variables, their cost and the number of
// objective evaluations per candidate
extern "C" int fuelmi_bspline_dev_optimize(fuelmi_bspline_dev* b, int max_eval, double* x_out, double* cost_out,
                                           int* evals_out) {
  return fuelmi_bspline_dev_optimize_timed(b, max_eval, -1.0, x_out, cost_out, evals_out);
}
extern "C" int fuelmi_bspline_dev_optimize_timed(fuelmi_bspline_dev* b, int max_eval, double max_time_s, double* x_out,
                                                 double* cost_out, int* evals_out) {
  ARGCHK(b && max_eval >= 1 && x_out && cost_out);
  fuelmi_map* m = b->map;
  HIPCHK(hipSetDevice(m->device));
  const BsplineArgs& A = b->a;
  const size_t C = (size_t)A.C, n = (size_t)A.nvar;
  const int npl = n <= 128 ? 2 : (n <= 256 ? 4 : 0);  // register-state kernel up to 256 variables
  const size_t lds_opt = npl ? b->lds_eval4 + 2 * n * sizeof(double)
                             : b->lds + ((6 + 2 * LBFGS_MEM) * n + 2 * LBFGS_MEM) * sizeof(double);
  if (lds_opt > 160 * 1024) {
    fuelmi_set_error("%d variables exceed the LDS budget of the device optimiser", (int)n);
    return FUELMI_ELIMIT;
  }
  if (!b->opt_x) {
    void* d = nullptr;
    if (lds_opt > 64 * 1024)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bspline_optimize),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_opt));
    HIPCHK(hipMalloc(&d, C * n * sizeof(double)));
    b->allocs.push_back(d);
    b->opt_x = (double*)d;
    HIPCHK(hipMalloc(&d, C * sizeof(double)));
    b->allocs.push_back(d);
    b->opt_cost = (double*)d;
    HIPCHK(hipMalloc(&d, C * sizeof(int)));
    b->allocs.push_back(d);
    b->opt_evals = (int*)d;
  }
  LbfgsArgs L;
  L.max_eval = max_eval;
  L.max_ticks = max_time_s > 0.0 ? (unsigned long long)(max_time_s * 1e8) + 1ull : 0ull;
  for (int k = 0; k < 3; ++k) {  // getBox(bmin, bmax); bmin += 0.1; bmax -= 0.1  (:174-178)
    L.box_lo[k] = m->cfg.box_min[k] + 0.1;
    L.box_hi[k] = m->cfg.box_max[k] - 0.1;
  }
  L.x_out = b->opt_x, L.cost_out = b->opt_cost, L.evals_out = b->opt_evals;
  {
    StageScope sc(m, FUELMI_K_BSPLINE);
    // four waves per candidate shorten ONE solve (the objective's terms side by side); a batch that fills the device
    // by itself (more candidates than CUs) is faster with one wave each: 1 024 solves 2.5 ms against 7.9
    const bool wide = A.C <= 256;
    if (npl == 2 && wide)
      k_bspline_optimize_r<2, 4><<<A.C, 256, lds_opt, m->stream>>>(m->g, m->dist, A, L);
    else if (npl == 4 && wide)
      k_bspline_optimize_r<4, 4><<<A.C, 256, lds_opt, m->stream>>>(m->g, m->dist, A, L);
    else if (npl == 2)
      k_bspline_optimize_r<2, 1><<<A.C, 64, lds_opt, m->stream>>>(m->g, m->dist, A, L);
    else if (npl == 4)
      k_bspline_optimize_r<4, 1><<<A.C, 64, lds_opt, m->stream>>>(m->g, m->dist, A, L);
    else
      k_bspline_optimize<<<A.C, 64, lds_opt, m->stream>>>(m->g, m->dist, A, L);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(x_out, b->opt_x, C * n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipMemcpyAsync(cost_out, b->opt_cost, C * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  std::vector<int> ev(C);
  HIPCHK(hipMemcpyAsync(ev.data(), b->opt_evals, C * sizeof(int), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  if (evals_out) memcpy(evals_out, ev.data(), C * sizeof(int));
  return FUELMI_OK;
}

extern "C" int fuelmi_bspline_dev_download(fuelmi_bspline_dev* b, double* cost, double* grad) {
  ARGCHK(b && cost && grad);
  HIPCHK(hipSetDevice(b->map->device));
  HIPCHK(hipMemcpyAsync(cost, b->a.cost, (size_t)b->a.C * sizeof(double), hipMemcpyDeviceToHost, b->map->stream));
  HIPCHK(hipMemcpyAsync(grad, b->a.grad, (size_t)b->a.C * b->a.nvar * sizeof(double), hipMemcpyDeviceToHost,
                        b->map->stream));
  HIPCHK(stream_wait(b->map->stream));  // (polls: a blocking synchronisation adds its wake-up latency to a 50 KB copy)
  return FUELMI_OK;
}

// One-shot calls (what the reference's callers make: combineCost / optimize() of ONE trajectory at a time,
// plan_manage/src/planner_manager.cpp:296-314) on a query slot of the map: the batch is packed into the slot's pinned
// block and read by the kernel in place, the kernel writes its results there, the caller polls the slot's event.  No
// hipMalloc / hipFree (round 3: ~10 allocations and a device-draining free per call, ~2.5 ms per solve), no copy
// engine, nothing on the map's own stream; concurrent callers get different slots.
// max_eval == 0: cost + gradient (out_a = cost [C], out_b = grad [C][nvar]); > 0: whole solves (out_a = cost_out [C],
// out_b = x_out [C][nvar], evals [C] or null).
static int bspline_oneshot(fuelmi_map* m, const fuelmi_bspline_cfg* cfg, const fuelmi_bspline_batch* in, int max_eval,
                           double max_time_s, double* out_a, double* out_b, int* evals) {
  ARGCHK(m && cfg && in && out_a && out_b);
  BsplineArgs A;
  {
    const int rca = bspline_args_init(cfg, in, A);
    if (rca) return rca;
  }
  HIPCHK(hipSetDevice(m->device));
  const int cf = A.cost_function;
  const size_t C = (size_t)A.C, n = (size_t)A.nvar;
  struct Piece {
    const void* src;
    size_t bytes;
    const void** dst;
  };
  const Piece pieces[] = {
      {in->x, C * n * sizeof(double), (const void**)&A.x},
      {in->pt_dist, C * sizeof(double), (const void**)&A.pt_dist},
      {in->knot_span, in->knot_span ? C * sizeof(double) : 0, (const void**)&A.knot_span},
      {in->time_lb, in->time_lb ? C * sizeof(double) : 0, (const void**)&A.time_lb},
      {in->start_state, (cf & FUELMI_COST_START) ? C * 9 * sizeof(double) : 0, (const void**)&A.start_state},
      {in->end_state, (cf & FUELMI_COST_END) ? C * 9 * sizeof(double) : 0, (const void**)&A.end_state},
      {in->guide_pts, ((cf & FUELMI_COST_GUIDE) && A.n_guide > 0) ? C * A.n_guide * 3 * sizeof(double) : 0, (const void**)&A.guide_pts},
      {in->waypoints, A.n_waypt > 0 ? C * A.n_waypt * 3 * sizeof(double) : 0, (const void**)&A.waypoints},
      {in->waypt_idx, A.n_waypt > 0 ? C * A.n_waypt * sizeof(int) : 0, (const void**)&A.waypt_idx},
      {in->view_pt, (cf & FUELMI_COST_VIEWCONS) ? C * 3 * sizeof(double) : 0, (const void**)&A.view_pt},
      {in->view_dir, (cf & FUELMI_COST_VIEWCONS) ? C * 3 * sizeof(double) : 0, (const void**)&A.view_dir},
      {in->view_idx, (cf & FUELMI_COST_VIEWCONS) ? C * sizeof(int) : 0, (const void**)&A.view_idx},
  };
  size_t in_bytes = 0;
  for (const Piece& p : pieces) in_bytes += (p.bytes + 15) & ~(size_t)15;
  const size_t out_bytes = ((C * sizeof(double) + 15) & ~(size_t)15) + ((C * n * sizeof(double) + 15) & ~(size_t)15) +
                           ((C * sizeof(int) + 15) & ~(size_t)15);
  const size_t lds_eval = ((size_t)A.N * 3 * 5 + EVAL_CONST) * sizeof(double);
  QuerySlotGuard q;
  {
    const int rcq = q.acquire(m, in_bytes + out_bytes);
    if (rcq) return rcq;
  }
  unsigned char* at = q.s->pin;
  for (const Piece& p : pieces) {
    *p.dst = nullptr;
    if (!p.bytes || !p.src) continue;
    memcpy(at, p.src, p.bytes);
    *p.dst = at;
    at += (p.bytes + 15) & ~(size_t)15;
  }
  double* o_a = reinterpret_cast<double*>(q.s->pin + in_bytes);
  double* o_b = reinterpret_cast<double*>(q.s->pin + in_bytes + ((C * sizeof(double) + 15) & ~(size_t)15));
  int* o_e = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(o_b) + ((C * n * sizeof(double) + 15) & ~(size_t)15));
  if (max_eval <= 0) {
    const size_t lds4 = lds_eval + ((size_t)A.N * 12 + 8) * sizeof(double);
    if (lds4 > 160 * 1024) {
      fuelmi_set_error("%d control points exceed the LDS budget", A.N);
      return FUELMI_ELIMIT;
    }
    if (lds4 > 64 * 1024)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bspline_cost_grad), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024));
    A.cost = o_a, A.grad = o_b;
    k_bspline_cost_grad<<<A.C, 256, lds4, q.s->st>>>(m->g, m->dist, A);
  } else {
    const int npl = n <= 128 ? 2 : (n <= 256 ? 4 : 0);
    const size_t lds_opt = npl ? lds_eval + ((size_t)A.N * 12 + 8) * sizeof(double) + 2 * n * sizeof(double)
                               : lds_eval + ((6 + 2 * LBFGS_MEM) * n + 2 * LBFGS_MEM) * sizeof(double);
    if (lds_opt > 160 * 1024) {
      fuelmi_set_error("%d variables exceed the LDS budget of the device optimiser", (int)n);
      return FUELMI_ELIMIT;
    }
    if (lds_opt > 64 * 1024 && !npl)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bspline_optimize), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_opt));
    LbfgsArgs L;
    L.max_eval = max_eval;
    L.max_ticks = max_time_s > 0.0 ? (unsigned long long)(max_time_s * 1e8) + 1ull : 0ull;
    for (int k = 0; k < 3; ++k) {
      L.box_lo[k] = m->cfg.box_min[k] + 0.1;
      L.box_hi[k] = m->cfg.box_max[k] - 0.1;
    }
    L.x_out = o_b, L.cost_out = o_a, L.evals_out = o_e;
    const bool wide = A.C <= 256;
    if (npl == 2 && wide)
      k_bspline_optimize_r<2, 4><<<A.C, 256, lds_opt, q.s->st>>>(m->g, m->dist, A, L);
    else if (npl == 4 && wide)
      k_bspline_optimize_r<4, 4><<<A.C, 256, lds_opt, q.s->st>>>(m->g, m->dist, A, L);
    else if (npl == 2)
      k_bspline_optimize_r<2, 1><<<A.C, 64, lds_opt, q.s->st>>>(m->g, m->dist, A, L);
    else if (npl == 4)
      k_bspline_optimize_r<4, 1><<<A.C, 64, lds_opt, q.s->st>>>(m->g, m->dist, A, L);
    else
      k_bspline_optimize<<<A.C, 64, lds_opt, q.s->st>>>(m->g, m->dist, A, L);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(q.finish());
  memcpy(out_a, o_a, C * sizeof(double));
  memcpy(out_b, o_b, C * n * sizeof(double));
  if (evals && max_eval > 0) memcpy(evals, o_e, C * sizeof(int));
  return FUELMI_OK;
}

extern "C" int fuelmi_bspline_cost_grad(fuelmi_map* m, const fuelmi_bspline_cfg* cfg,
                                        const fuelmi_bspline_batch* batch, double* cost, double* grad) {
  ARGCHK(cost && grad);
  return bspline_oneshot(m, cfg, batch, 0, -1.0, cost, grad, nullptr);
}
extern "C" int fuelmi_bspline_optimize(fuelmi_map* m, const fuelmi_bspline_cfg* cfg, const fuelmi_bspline_batch* batch,
                                       int max_eval, double max_time_s, double* x_out, double* cost_out, int* evals_out) {
  ARGCHK(max_eval >= 1 && x_out && cost_out);
  return bspline_oneshot(m, cfg, batch, max_eval, max_time_s, cost_out, x_out, evals_out);
}

// ---- spline glue entry points ----
namespace {
int fit_launch(fuelmi_map* m, const FitArgs& F, hipStream_t st = nullptr) {
  const size_t lds = fit_lds(F.K, F.degree);
  if (lds > 160 * 1024) {
    fuelmi_set_error("%d samples exceed the LDS budget of the spline fit", F.K);
    return FUELMI_ELIMIT;
  }
  if (lds > 64 * 1024)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bspline_fit), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  k_bspline_fit<<<F.C, 64, lds, st ? st : m->stream>>>(F);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}
}  // namespace

extern "C" int fuelmi_bspline_parameterize(fuelmi_map* m, int n_traj, int n_points, int degree, const double* ts,
                                           const double* points, const double* derivs, double* ctrl) {
  ARGCHK(m && ts && points && derivs && ctrl);
  ARGCHK(n_traj >= 1 && n_points >= 2 && degree >= 3 && degree <= 5);
  for (int c = 0; c < n_traj; ++c)
    if (!(ts[c] > 0)) {  // "[B-spline]:time step error." (:181-184)
      fuelmi_set_error("fuelmi_bspline_parameterize: time step of candidate %d is not positive", c);
      return FUELMI_EINVAL;
    }
  HIPCHK(hipSetDevice(m->device));
  const size_t C = (size_t)n_traj, K = (size_t)n_points, n = K + degree - 1;
  const size_t b_ts = C * sizeof(double), b_pts = C * K * 3 * sizeof(double), b_der = C * 12 * sizeof(double),
               b_ctrl = C * n * 3 * sizeof(double);
  QuerySlotGuard q;  // (inputs read and results written in the slot's pinned block: no allocation, no copies)
  {
    const int rcq = q.acquire(m, b_ts + b_pts + b_der + b_ctrl);
    if (rcq) return rcq;
  }
  unsigned char* d = q.s->pin;
  memcpy(d, ts, b_ts);
  memcpy(d + b_ts, points, b_pts);
  memcpy(d + b_ts + b_pts, derivs, b_der);
  FitArgs F;
  memset(&F, 0, sizeof(F));
  F.C = n_traj, F.K = n_points, F.degree = degree;
  F.ts = reinterpret_cast<double*>(d);
  F.points = reinterpret_cast<double*>(d + b_ts);
  F.derivs = reinterpret_cast<double*>(d + b_ts + b_pts);
  F.ctrl = reinterpret_cast<double*>(d + b_ts + b_pts + b_der);
  F.stride = (long)(n * 3);
  const int rc = fit_launch(m, F, q.s->st);
  if (rc) return rc;
  HIPCHK(q.finish());
  memcpy(ctrl, F.ctrl, b_ctrl);
  return FUELMI_OK;
}

extern "C" int fuelmi_bspline_boundary_states(fuelmi_map* m, int n_traj, int n_ctrl, int degree, const double* ts,
                                              const double* ctrl, int ks, int ke, double* start, double* end) {
  ARGCHK(m && ts && ctrl && start && end);
  ARGCHK(n_traj >= 1 && degree >= 1 && degree <= 5 && n_ctrl > degree);
  ARGCHK(ks >= 0 && ke >= 0 && ks <= degree && ke <= degree);
  HIPCHK(hipSetDevice(m->device));
  const size_t C = (size_t)n_traj, n = (size_t)n_ctrl;
  const size_t b_ts = C * sizeof(double), b_ctrl = C * n * 3 * sizeof(double), b_s = C * (ks + 1) * 3 * sizeof(double),
               b_e = C * (ke + 1) * 3 * sizeof(double);
  const size_t lds = (3 * n + n + degree + 1) * sizeof(double);
  if (lds > 64 * 1024) {
    fuelmi_set_error("%d control points exceed the LDS budget", n_ctrl);
    return FUELMI_ELIMIT;
  }
  QuerySlotGuard q;
  {
    const int rcq = q.acquire(m, b_ts + b_ctrl + b_s + b_e);
    if (rcq) return rcq;
  }
  unsigned char* d = q.s->pin;
  memcpy(d, ts, b_ts);
  memcpy(d + b_ts, ctrl, b_ctrl);
  double* d_s = reinterpret_cast<double*>(d + b_ts + b_ctrl);
  double* d_e = reinterpret_cast<double*>(d + b_ts + b_ctrl + b_s);
  k_bspline_boundary<<<n_traj, 64, lds, q.s->st>>>(n_ctrl, degree, reinterpret_cast<double*>(d),
                                                    reinterpret_cast<double*>(d + b_ts), ks, ke, d_s, d_e);
  HIPCHK(hipGetLastError());
  HIPCHK(q.finish());
  memcpy(start, d_s, b_s);
  memcpy(end, d_e, b_e);
  return FUELMI_OK;
}

// planner glue on the device: samples -> control points + knot span -> getBoundaryStates(2, 0) ->
// setBoundaryStates + pt_dist_, written into the batch's own device state (asynchronous; the next
// _eval / _optimize sees it)
extern "C" int fuelmi_bspline_dev_load_samples(fuelmi_bspline_dev* b, int n_points, const double* ts,
                                               const double* points, const double* derivs) {
  ARGCHK(b && ts && points && derivs);
  BsplineArgs& A = b->a;
  const int degree = A.cfg.bspline_degree;
  ARGCHK(A.dim == 3 && degree >= 3 && degree <= 5 && n_points >= 2 && n_points + degree - 1 == A.N);
  fuelmi_map* m = b->map;
  HIPCHK(hipSetDevice(m->device));
  const size_t C = (size_t)A.C, K = (size_t)n_points;
  const size_t b_ts = C * sizeof(double), b_pts = C * K * 3 * sizeof(double), b_der = C * 12 * sizeof(double);
  if (b_ts + b_pts + b_der > b->fit_cap) {
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, b_ts + b_pts + b_der));
    b->allocs.push_back(d);
    b->fit_in = static_cast<double*>(d);
    b->fit_cap = b_ts + b_pts + b_der;
  }
  unsigned char* d = reinterpret_cast<unsigned char*>(b->fit_in);
  HIPCHK(hipMemcpyAsync(d, ts, b_ts, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(d + b_ts, points, b_pts, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipMemcpyAsync(d + b_ts + b_pts, derivs, b_der, hipMemcpyHostToDevice, m->stream));
  FitArgs F;
  memset(&F, 0, sizeof(F));
  F.C = A.C, F.K = n_points, F.degree = degree;
  F.ts = reinterpret_cast<double*>(d);
  F.points = reinterpret_cast<double*>(d + b_ts);
  F.derivs = reinterpret_cast<double*>(d + b_ts + b_pts);
  F.ctrl = const_cast<double*>(A.x);
  F.stride = A.nvar;
  F.write_dt = (A.cost_function & FUELMI_COST_MINTIME) ? 1 : 0;
  F.knot_span = const_cast<double*>(A.knot_span);
  F.pt_dist = const_cast<double*>(A.pt_dist);
  F.start_state = const_cast<double*>(A.start_state);
  F.end_state = const_cast<double*>(A.end_state);
  StageScope sc(m, FUELMI_K_BSPLINE);
  return fit_launch(m, F);
}
