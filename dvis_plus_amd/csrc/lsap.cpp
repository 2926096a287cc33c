// Host-side linear sum assignment for the tracker's query matching (Noiser.match_embds,
// dvis_Plus/noiser.py:43-56 calls scipy.optimize.linear_sum_assignment(C.T)[1]).
//
// scipy (pinned scipy==1.5.4 in the reference's requirements.txt; 1.15.3 in this image) is an un-vendored
// third-party dependency; its solver is the shortest-augmenting-path algorithm of
//   D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016
// (a Jonker-Volgenant variant with dual variables u, v).  This file restates that published algorithm.
// To return the SAME permutation on degenerate (tied) cost matrices the scan conventions that decide
// ties are kept: unscanned columns are visited from a list initialised in descending column order with
// swap-with-last removal, and among equal shortest-path costs an unassigned column is preferred.
// Index parity is pinned by tests/golden/g5_match.npz (random, permuted, duplicated-row and zero-vector cases).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <thread>
#include <vector>

#include "dvis_common.h"

// Shortest augmenting paths with duals.  u (nr) / v (nc), when given, receive the final dual variables: feasible
// (cost[i][j] - u[i] - v[j] >= 0 up to rounding) and tight on the assignment — the optimality certificate the chain below
// uses to prove an optimum unique.
static int lsap_core(const double *cost, int nr, int nc, int64_t *col4row_out, double *u_out, double *v_out) {
  for (long long k = 0; k < (long long)nr * nc; ++k)
    DVIS_REQUIRE(!(std::isnan(cost[k]) || cost[k] == -std::numeric_limits<double>::infinity()),
                 "lsap: cost matrix contains nan or -inf");
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<double> u(nr, 0.0), v(nc, 0.0), dist(nc);
  std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), todo(nc);
  std::vector<char> rowSeen(nr), colSeen(nc);

  for (int cur = 0; cur < nr; ++cur) {
    // ---- shortest augmenting path from row `cur` to an unassigned column
    std::fill(rowSeen.begin(), rowSeen.end(), 0);
    std::fill(colSeen.begin(), colSeen.end(), 0);
    std::fill(dist.begin(), dist.end(), INF);
    int ntodo = nc;
    for (int t = 0; t < nc; ++t) todo[t] = nc - 1 - t;
    double minVal = 0.0;
    int i = cur, sink = -1;
    while (sink < 0) {
      int pick = -1;
      double lowest = INF;
      rowSeen[i] = 1;
      const double *ci = cost + (long long)i * nc;
      for (int t = 0; t < ntodo; ++t) {
        const int j = todo[t];
        const double r = minVal + ci[j] - u[i] - v[j];
        if (r < dist[j]) {
          dist[j] = r;
          path[j] = i;
        }
        if (dist[j] < lowest || (dist[j] == lowest && row4col[j] < 0)) {
          lowest = dist[j];
          pick = t;
        }
      }
      minVal = lowest;
      if (minVal == INF) {
        dvis_set_error("lsap: cost matrix is infeasible");
        return DVIS_E_ARG;
      }
      const int j = todo[pick];
      if (row4col[j] < 0)
        sink = j;
      else
        i = row4col[j];
      colSeen[j] = 1;
      todo[pick] = todo[--ntodo];
    }
    // ---- dual update
    u[cur] += minVal;
    for (int r = 0; r < nr; ++r)
      if (rowSeen[r] && r != cur) u[r] += minVal - dist[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (colSeen[j]) v[j] -= minVal - dist[j];
    // ---- flip the assignments along the path
    int j = sink;
    while (true) {
      const int r = path[j];
      row4col[j] = r;
      const int prev = col4row[r];
      col4row[r] = j;
      j = prev;
      if (r == cur) break;
    }
  }
  for (int r = 0; r < nr; ++r) col4row_out[r] = col4row[r];
  if (u_out) std::copy(u.begin(), u.end(), u_out);
  if (v_out) std::copy(v.begin(), v.end(), v_out);
  return DVIS_OK;
}

DVIS_EXPORT int dvis_lsap_solve(const double *cost, int nr, int nc, int64_t *col4row_out) {
  DVIS_REQUIRE(cost && col4row_out, "lsap: null pointer");
  DVIS_REQUIRE(nr >= 0 && nc >= 0 && nr <= nc, "lsap: need 0 <= nr <= nc (got %d x %d)", nr, nc);
  return lsap_core(cost, nr, nc, col4row_out, nullptr, nullptr);
}

// The whole per-clip matching recurrence of the referring tracker on the host, in one call.
//   cost (T, Q, Q) fp32: cost[i][c][r'] = 1 - cos(cur_i[c], cur_{i-1}[r'])   (frame 0: vs the carried-over reference
//   embeddings, or vs itself at the start of a video) — computed on the GPU for all frames by ONE batched GEMM.
// Reference (dvis_Plus/tracker.py:283-291, noiser.py:43-56): frame i is matched against the RE-ORDERED embeddings of
// frame i-1, last_frame_embeds = cur_{i-1}[idx_{i-1}]; re-ordering rows of the reference only permutes the columns
// of the similarity matrix, so  C_i[c][r] = cost[i][c][idx_{i-1}[r]]  and the assignment is solved on C_i^T
// (rows = reference slots) exactly like linear_sum_assignment(C.transpose(0, 1))[1].  NaN costs become 0 (noiser.py:52).
//
// The frames' problems differ from their "canonical" forms (reference slots in their original order) only by a permutation
// of the ROWS, which the solver's answer follows — unless the optimum is not unique, where the scan order decides between
// equal-cost assignments (and scipy's choice must be reproduced).  So the chain is solved in two passes:
//   1. every frame's canonical problem on its own host thread, with a uniqueness certificate from the dual variables:
//      no alternating cycle of edges with reduced cost cost - u - v <= 1e-9 exists, i.e. any other assignment is worse by
//      more than 1e-9 while the solver's rounding is ~1e-13 for costs in [0, 2] — whatever the row order, the sequential
//      solve finds this assignment;
//   2. the T-step composition in order: idx_i[r] = sigma_i[idx_{i-1}[r]] where certified, else the exact sequential solve on
//      the permuted matrix (ties: duplicate queries, zero vectors — tests/golden/g5_match.npz).
// Same indices as the frame-by-frame loop, 1.7 - 7.8 ms -> a few hundred us per 30-frame clip on the host.
// DVIS_MATCH_THREADS: 1 = the sequential loop.
namespace {

struct ChainFrame {
  std::vector<int64_t> sigma;
  bool certified = false;
  int rc = DVIS_OK;
};

void canonical_cost(const float *ci, int Q, const int64_t *prev, std::vector<double> &c) {
  for (int r = 0; r < Q; ++r) {
    const int64_t src = prev ? prev[r] : r;
    for (int q = 0; q < Q; ++q) {
      const float x = ci[(size_t)q * Q + src];
      c[(size_t)r * Q + q] = std::isnan(x) ? 0.0 : (double)x;
    }
  }
}

void solve_canonical(const float *ci, int Q, ChainFrame &f) {
  std::vector<double> c((size_t)Q * Q), u(Q), v(Q);
  canonical_cost(ci, Q, nullptr, c);
  f.sigma.resize(Q);
  f.rc = lsap_core(c.data(), Q, Q, f.sigma.data(), u.data(), v.data());
  if (f.rc != DVIS_OK) return;
  // Another optimal (or within-tolerance) assignment differs from sigma by alternating cycles: rows r1 -> r2 -> ... -> r1
  // where r_k takes the column of r_{k+1}.  Its extra cost is the sum of the reduced costs of those edges (all >= 0), so the
  // optimum is unique with margin eps iff the digraph { r -> r' : reduced cost of (r, sigma[r']) <= eps } has no cycle
  // (tight edges OUTSIDE cycles are normal: the duals of a unique optimum are usually degenerate).  Kahn's algorithm.
  const double eps = 1e-9;
  std::vector<char> tight((size_t)Q * Q, 0);
  std::vector<int> indeg(Q, 0), stack;
  for (int r = 0; r < Q; ++r) {
    const double *cr = c.data() + (size_t)r * Q;
    for (int r2 = 0; r2 < Q; ++r2) {
      const int64_t q = f.sigma[r2];
      if (r2 != r && !(cr[q] - u[r] - v[q] > eps)) {
        tight[(size_t)r * Q + r2] = 1;
        ++indeg[r2];
      }
    }
  }
  for (int r = 0; r < Q; ++r)
    if (indeg[r] == 0) stack.push_back(r);
  int removed = 0;
  while (!stack.empty()) {
    const int r = stack.back();
    stack.pop_back();
    ++removed;
    for (int r2 = 0; r2 < Q; ++r2)
      if (tight[(size_t)r * Q + r2] && --indeg[r2] == 0) stack.push_back(r2);
  }
  const bool unique = removed == Q;
  f.certified = unique;
}

}  // namespace

DVIS_EXPORT int dvis_match_chain(const float *cost, int T, int Q, int64_t *indices) {
  DVIS_REQUIRE(cost && indices, "match_chain: null pointer");
  DVIS_REQUIRE(T >= 0 && Q > 0, "match_chain: bad sizes");
  static const int max_threads = []() {
    const char *e = getenv("DVIS_MATCH_THREADS");
    if (e) return std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::min(16u, std::max(1u, hw));
  }();
  const int nthreads = std::min(max_threads, T);
  std::vector<ChainFrame> frames(nthreads > 1 ? T : 0);
  if (nthreads > 1) {
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t]() {
        for (int i = t; i < T; i += nthreads) solve_canonical(cost + (size_t)i * Q * Q, Q, frames[i]);
      });
    for (auto &th : pool) th.join();
  }
  std::vector<double> c((size_t)Q * Q);
  for (int i = 0; i < T; ++i) {
    const float *ci = cost + (size_t)i * Q * Q;
    const int64_t *prev = i == 0 ? nullptr : indices + (size_t)(i - 1) * Q;
    int64_t *out = indices + (size_t)i * Q;
    if (nthreads > 1 && frames[i].rc == DVIS_OK && frames[i].certified) {
      for (int r = 0; r < Q; ++r) out[r] = frames[i].sigma[prev ? prev[r] : r];
      continue;
    }
    canonical_cost(ci, Q, prev, c);
    int rc = dvis_lsap_solve(c.data(), Q, Q, out);
    if (rc != DVIS_OK) return rc;
  }
  return DVIS_OK;
}
