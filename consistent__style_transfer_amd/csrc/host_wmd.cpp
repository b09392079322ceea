// Host side of the pretrain stage's content-distance labels (reference: src/wmd.py:31-45 `cal_wmd_label`, called per batch from
// src/loader.py:60): the Word Mover's Distance between the two noised copies of every sentence -- 256 exact transportation problems
// per batch.  The reference gets them from gensim's `wmdistance` (pyemd, third-party C++); round 2 of this build restated the
// algorithm on numpy + scipy's HiGHS LP (wmd.py) at 281 labels / s / core, ~100x short of what the GPU's pretrain step consumes.
// This file is the same algorithm as one C entry point over a whole batch:
//
//   per pair:  drop out-of-vocabulary tokens; either side empty -> inf; one distinct token in total -> 0;
//              cost[i][j] = ||v_i - v_j||_2 between doc-1 and doc-2 tokens (vectors L2-normalised by the caller: wmd.py:54);
//              all costs 0 -> inf; nBOW weights = count / (in-vocabulary length);
//              distance = min sum F_ij cost_ij  s.t.  F >= 0, F 1 = nbow_1, F^T 1 = nbow_2
//   label   :  an empty id list -> max(len_1, len_2) (wmd.py:37-38); distance inf -> (len_1 + len_2) / 2 (wmd.py:41-42); else the distance.
//
// The transportation problem (<= max_len distinct tokens a side: 18 x 18 on Yelp, 30 x 30 on the book corpus) is solved EXACTLY by
// successive shortest augmenting paths on the bipartite residual graph with node potentials (dense Dijkstra over n + m nodes; reduced
// costs stay >= 0, every augmentation exhausts a supply, a demand or a back edge).  Plain C++17, no GPU, no third-party code; built by
// consistent__style_transfer_amd/build.py into libcst_host.so next to libcst_hip.so.  The C ABI is include/cst_host.h.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

// min-cost transport of supplies a[n] to demands b[m] (equal total mass), cost[n*m] row-major.  Returns the optimal cost.
double transport_cost(int n, int m, const double* a, const double* b, const double* cost, std::vector<double>& flow,
                      std::vector<double>& scratch) {
    if (n == 1) { double s = 0; for (int j = 0; j < m; ++j) s += cost[j] * b[j]; return s; }
    if (m == 1) { double s = 0; for (int i = 0; i < n; ++i) s += cost[i] * a[i]; return s; }
    const int N = n + m;
    flow.assign((size_t)n * m, 0.0);
    scratch.assign((size_t)2 * N + n + m, 0.0);
    double* pot = scratch.data();               // node potentials [N]
    double* dist = pot + N;                     // [N]
    double* sup = dist + N;                     // remaining supply [n]
    double* dem = sup + n;                      // remaining demand [m]
    std::vector<int> prev(N), done(N);
    double total = 0.0;
    for (int i = 0; i < n; ++i) { sup[i] = a[i]; total += a[i]; }
    for (int j = 0; j < m; ++j) dem[j] = b[j];
    const double eps = 1e-13 * (total > 0 ? total : 1.0);
    // potentials start at 0: all forward reduced costs = cost >= 0
    for (int it = 0; it < 4 * N * N + 16; ++it) {
        // Dijkstra from every source with supply left; nodes 0..n-1 = sources, n..N-1 = sinks
        bool any = false;
        for (int v = 0; v < N; ++v) { dist[v] = kInf; prev[v] = -1; done[v] = 0; }
        for (int i = 0; i < n; ++i) if (sup[i] > eps) { dist[i] = 0.0; any = true; }
        if (!any) break;
        for (int round = 0; round < N; ++round) {
            int u = -1; double du = kInf;
            for (int v = 0; v < N; ++v) if (!done[v] && dist[v] < du) { du = dist[v]; u = v; }
            if (u < 0) break;
            done[u] = 1;
            if (u < n) {                          // source i -> every sink j (capacity unbounded)
                const double* c = cost + (size_t)u * m;
                for (int j = 0; j < m; ++j) {
                    const int v = n + j;
                    if (done[v]) continue;
                    double rc = c[j] + pot[u] - pot[v];
                    if (rc < 0) rc = 0;           // rounding only: reduced costs are >= 0 in exact arithmetic
                    if (du + rc < dist[v]) { dist[v] = du + rc; prev[v] = u; }
                }
            } else {                              // sink j -> source i along a back edge with flow
                const int j = u - n;
                for (int i = 0; i < n; ++i) {
                    if (done[i] || flow[(size_t)i * m + j] <= eps) continue;
                    double rc = -cost[(size_t)i * m + j] + pot[u] - pot[i];
                    if (rc < 0) rc = 0;
                    if (du + rc < dist[i]) { dist[i] = du + rc; prev[i] = u; }
                }
            }
        }
        // nearest sink with demand left
        int t = -1; double dt = kInf;
        for (int j = 0; j < m; ++j) if (dem[j] > eps && dist[n + j] < dt) { dt = dist[n + j]; t = n + j; }
        if (t < 0) break;
        for (int v = 0; v < N; ++v) pot[v] += (dist[v] < dt ? dist[v] : dt);      // keeps every residual reduced cost >= 0
        // bottleneck along the path
        double amt = dem[t - n];
        int v = t;
        while (prev[v] >= 0) {
            const int u = prev[v];
            if (u >= n) amt = std::min(amt, flow[(size_t)v * m + (u - n)]);     // back edge sink u -> source v
            v = u;
        }
        amt = std::min(amt, sup[v]);
        // augment
        dem[t - n] -= amt;
        sup[v] -= amt;
        v = t;
        while (prev[v] >= 0) {
            const int u = prev[v];
            if (u < n) flow[(size_t)u * m + (v - n)] += amt;
            else flow[(size_t)v * m + (u - n)] -= amt;
            v = u;
        }
    }
    double obj = 0.0;
    for (size_t k = 0; k < (size_t)n * m; ++k) obj += flow[k] * cost[k];
    return obj;
}

struct PairWork {
    std::vector<int> t1, t2, u1, u2;      // in-vocabulary rows of both documents; their distinct rows
    std::vector<double> w1, w2, cost, flow, scratch;
};

double wm_distance(const int32_t* x1, int n1, const int32_t* x2, int n2, const int32_t* row_of_id, int n_ids,
                   const double* vec, int dim, PairWork& w) {
    auto rows = [&](const int32_t* x, int n, std::vector<int>& out) {
        out.clear();
        for (int k = 0; k < n; ++k) {
            const int32_t id = x[k];
            if (id >= 0 && id < n_ids && row_of_id[id] >= 0) out.push_back(row_of_id[id]);
        }
    };
    rows(x1, n1, w.t1);
    rows(x2, n2, w.t2);
    if (w.t1.empty() || w.t2.empty()) return kInf;
    auto distinct = [](const std::vector<int>& t, std::vector<int>& u) { u = t; std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end()); };
    distinct(w.t1, w.u1);
    distinct(w.t2, w.u2);
    if (w.u1.size() == 1 && w.u2.size() == 1 && w.u1[0] == w.u2[0]) return 0.0;      // one distinct token in total
    const int n = (int)w.u1.size(), m = (int)w.u2.size();
    w.cost.resize((size_t)n * m);
    double csum = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* a = vec + (size_t)w.u1[i] * dim;
        for (int j = 0; j < m; ++j) {
            double d = 0.0;
            if (w.u1[i] != w.u2[j]) {
                const double* b = vec + (size_t)w.u2[j] * dim;
                double s = 0.0;
                for (int k = 0; k < dim; ++k) { const double t = a[k] - b[k]; s += t * t; }
                d = std::sqrt(s);
            }
            w.cost[(size_t)i * m + j] = d;
            csum += d;
        }
    }
    if (csum == 0.0) return kInf;
    w.w1.assign(n, 0.0);
    w.w2.assign(m, 0.0);
    for (int r : w.t1) w.w1[std::lower_bound(w.u1.begin(), w.u1.end(), r) - w.u1.begin()] += 1.0 / (double)w.t1.size();
    for (int r : w.t2) w.w2[std::lower_bound(w.u2.begin(), w.u2.end(), r) - w.u2.begin()] += 1.0 / (double)w.t2.size();
    return transport_cost(n, m, w.w1.data(), w.w2.data(), w.cost.data(), w.flow, w.scratch);
}

}  // namespace

extern "C" {

// exact earth mover's distance of two histograms of equal mass: a[n], b[m] > 0, cost[n*m] row-major.  0 on success.
int cst_host_emd(int n, int m, const double* a, const double* b, const double* cost, double* out) {
    if (n <= 0 || m <= 0 || !a || !b || !cost || !out) return 1;
    std::vector<double> flow, scratch;
    *out = transport_cost(n, m, a, b, cost, flow, scratch);
    return 0;
}

// labels[p] for pairs p in [p_lo, p_hi) of n_pairs sentence pairs given as ragged int32 id lists (ids1 / off1[n_pairs + 1], ids2 / off2);
// row_of_id[n_ids]: vocabulary id -> row of `vectors` [n_rows, dim] (L2-normalised, float64) or -1 (out of the word-vector vocabulary).
// Pairs outside [p_lo, p_hi) are left untouched (a data-parallel rank computes its own rows only).  nthreads >= 1.  0 on success.
int cst_host_wmd_labels(const int32_t* ids1, const int64_t* off1, const int32_t* ids2, const int64_t* off2, int64_t n_pairs,
                        int64_t p_lo, int64_t p_hi, const int32_t* row_of_id, int32_t n_ids, const double* vectors, int32_t dim,
                        int32_t nthreads, double* labels) {
    if (!ids1 || !off1 || !ids2 || !off2 || !row_of_id || !vectors || !labels || n_pairs < 0 || dim <= 0) return 1;
    if (p_lo < 0) p_lo = 0;
    if (p_hi > n_pairs) p_hi = n_pairs;
    if (nthreads < 1) nthreads = 1;
    auto run = [&](int64_t lo, int64_t hi) {
        PairWork w;
        for (int64_t p = lo; p < hi; ++p) {
            const int n1 = (int)(off1[p + 1] - off1[p]), n2 = (int)(off2[p + 1] - off2[p]);
            if (n1 == 0 || n2 == 0) { labels[p] = (double)std::max(n1, n2); continue; }                 // wmd.py:37-38
            const double d = wm_distance(ids1 + off1[p], n1, ids2 + off2[p], n2, row_of_id, n_ids, vectors, dim, w);
            labels[p] = std::isinf(d) ? (double)(n1 + n2) / 2.0 : d;                                   // wmd.py:41-44
        }
    };
    const int64_t n = p_hi - p_lo;
    if (nthreads == 1 || n < 2 * nthreads) { run(p_lo, p_hi); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t lo = p_lo + n * t / nthreads, hi = p_lo + n * (t + 1) / nthreads;
        th.emplace_back(run, lo, hi);
    }
    for (auto& t : th) t.join();
    return 0;
}

int cst_host_abi_version() { return 1; }

}  // extern "C"
