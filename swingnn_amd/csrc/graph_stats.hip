// Binarised-graph statistics of the reference's evaluation on the device (evaluation/stats.py:26-175,
// evaluation/mmd.py:82-163): degree histogram, clustering-coefficient histogram, normalised Laplacian (for the
// spectral statistic), the numpy-compatible uniform histogram, and the Gaussian-TV MMD between two sets of
// histograms.  The reference does this with networkx + Python double loops on the host after sampling.
//
// Integer work on bitsets (HBM / LDS bound, exact): an adjacency row of N <= 1024 nodes is N/32 words; degree =
// popcount(row), triangles at i = 1/2 sum_{j in N(i)} popcount(row_i & row_j).  Floating-point parts are fp64 with the
// operation order of numpy (no fused multiply-add where numpy has none), so histograms are bit-compatible.
#include "common.h"

namespace {

constexpr int MAXW = 32;  // words per adjacency row: N <= 1024

// one workgroup per graph: bitset rows -> LDS-free passes over global bitsets (N*W words, L2-resident)
__global__ __launch_bounds__(256) void k_graph_bits(int N, int W, const float* __restrict__ adj, uint32_t* __restrict__ bits) {
  const int b = blockIdx.y;
  const float* a = adj + (size_t)b * N * N;
  uint32_t* out = bits + (size_t)b * N * W;
  // word (i, w) = bits j = 32w .. 32w+31 of row i: (a_ij != 0 or a_ji != 0) and i != j  (stats.py:291-305: symmetric,
  // no self loops)
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * W; idx += gridDim.x * 256) {
    const int i = idx / W, w = idx - i * W;
    uint32_t m = 0;
    for (int k = 0; k < 32; ++k) {
      const int j = 32 * w + k;
      if (j < N && j != i && (a[(size_t)i * N + j] != 0.f || a[(size_t)j * N + i] != 0.f)) m |= 1u << k;
    }
    out[idx] = m;
  }
}

// np.histogram fast path for uniform bins (numpy/lib/histograms.py): the final bin of v is the k with
// edges[k] <= v < edges[k+1] (last bin closed), edges = first + arange(bins+1) * step with the last edge set to `last`
__device__ __forceinline__ double np_edge(int k, int bins, double first, double last, double step) {
  return k == bins ? last : __dadd_rn(__dmul_rn((double)k, step), first);
}
__device__ __forceinline__ int np_bin(double v, int bins, double first, double last) {
  if (!(v >= first && v <= last)) return -1;
  const double step = __ddiv_rn(__dadd_rn(last, -first), (double)bins);
  const double norm = __ddiv_rn((double)bins, __dadd_rn(last, -first));
  int idx = (int)__dmul_rn(__dadd_rn(v, -first), norm);
  if (idx == bins) idx -= 1;
  if (v < np_edge(idx, bins, first, last, step)) idx -= 1;
  else if (idx != bins - 1 && v >= np_edge(idx + 1, bins, first, last, step)) idx += 1;
  return idx;
}

// one workgroup per graph: degrees, triangles -> degree histogram [N] (nx.degree_histogram over the graph without its
// isolated nodes), clustering histogram [cbins] over [0,1], and optionally the normalised Laplacian [N,N] (fp64) with
// the sentinel `iso` on the diagonal of removed (isolated) nodes so that their eigenvalues fall outside the histogram
__global__ __launch_bounds__(256) void k_graph_deg_clu(int N, int W, int cbins, const uint32_t* __restrict__ bits,
                                                       double* __restrict__ deg_hist, double* __restrict__ clu_hist,
                                                       double* __restrict__ lap, double iso) {
  extern __shared__ int sm[];
  int* s_deg = sm;              // [N]
  int* s_hd = sm + N;           // [N] degree histogram
  int* s_hc = s_hd + N;         // [cbins]
  int* s_any = s_hc + cbins;    // [1]
  const int b = blockIdx.x;
  const uint32_t* rows = bits + (size_t)b * N * W;
  for (int i = threadIdx.x; i < N; i += 256) s_hd[i] = 0;
  for (int i = threadIdx.x; i < cbins; i += 256) s_hc[i] = 0;
  if (threadIdx.x == 0) *s_any = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) {
    int d = 0;
    for (int w = 0; w < W; ++w) d += __popc(rows[(size_t)i * W + w]);
    s_deg[i] = d;
    if (d > 0) *s_any = 1;  // benign race: every writer stores 1
  }
  __syncthreads();
  const bool any = *s_any != 0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const int d = s_deg[i];
    // adjs_to_graphs drops isolated nodes; a graph with no edge at all is kept as ONE node (stats.py:299-303)
    const bool kept = d > 0 || (!any && i == 0);
    if (!kept) continue;
    uint32_t mine[MAXW];
    for (int w = 0; w < W; ++w) mine[w] = rows[(size_t)i * W + w];
    long tri2 = 0;  // closed walks of length 3 through i = 2 * triangles at i
    for (int w = 0; w < W; ++w) {
      uint32_t m = mine[w];
      while (m) {
        const int j = 32 * w + __ffs(m) - 1;
        m &= m - 1;
        for (int u = 0; u < W; ++u) tri2 += __popc(mine[u] & rows[(size_t)j * W + u]);
      }
    }
    atomicAdd(&s_hd[d], 1);
    const double tri = (double)tri2 / 2.0, denom = (double)d * (double)(d - 1) / 2.0;
    const double c = denom > 0.0 ? tri / denom : 0.0;  // nx.clustering: 0 for degree < 2
    const int k = np_bin(c, cbins, 0.0, 1.0);
    if (k >= 0) atomicAdd(&s_hc[k], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) deg_hist[(size_t)b * N + i] = (double)s_hd[i];
  for (int i = threadIdx.x; i < cbins; i += 256) clu_hist[(size_t)b * cbins + i] = (double)s_hc[i];
  if (lap) {
    // nx.normalized_laplacian_matrix (stats.py:128-131): L = I - D^-1/2 A D^-1/2 on the kept nodes
    double* L = lap + (size_t)b * N * N;
    for (int idx = threadIdx.x; idx < N * N; idx += 256) {
      const int i = idx / N, j = idx - i * N;
      const int di = s_deg[i], dj = s_deg[j];
      double v = 0.0;
      if (i == j) v = di > 0 ? 1.0 : ((!any && i == 0) ? 0.0 : iso);
      else if ((rows[(size_t)i * W + (j >> 5)] >> (j & 31)) & 1u) v = -(1.0 / sqrt((double)di)) * (1.0 / sqrt((double)dj));
      L[idx] = v;
    }
  }
}

// rows of `vals` [B, n] -> numpy histogram over [first, last] with `bins` uniform bins, as a pmf (count / total)
// `snap` > 0: a value at most `snap` outside the range is moved onto its end (an eigenvalue of a normalised Laplacian
// lies in [0, 2]; a solver may return 2 + 4e-16, which np.histogram -- and with snap = 0 this kernel -- drops)
__global__ __launch_bounds__(256) void k_hist_pmf(int n, int bins, const double* __restrict__ vals, double first,
                                                  double last, double snap, double* __restrict__ out) {
  extern __shared__ int sh[];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i <= bins; i += 256) sh[i] = 0;  // sh[bins] = total
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    double v = vals[(size_t)b * n + i];
    if (v > last && v <= last + snap) v = last;
    if (v < first && v >= first - snap) v = first;
    const int k = np_bin(v, bins, first, last);
    if (k >= 0) { atomicAdd(&sh[k], 1); atomicAdd(&sh[bins], 1); }
  }
  __syncthreads();
  const double tot = (double)sh[bins];
  for (int i = threadIdx.x; i < bins; i += 256) out[(size_t)b * bins + i] = (double)sh[i] / tot;
}

// Gaussian-TV kernel matrix (mmd.py:82-95) over Z = [X; Y] (n rows of length L, each first normalised to sum 1 when
// its sum is not 0, mmd.py:146-149), then MMD^2 = mean k(X,X) + mean k(Y,Y) - 2 mean k(X,Y) (mmd.py:151-163).
// Block i handles row i against all j; per-row partial sums go to scratch and are folded in row order by one thread:
// deterministic.
__global__ __launch_bounds__(64) void k_mmd_row_sums(int n1, int n2, int L, const double* __restrict__ X,
                                                     const double* __restrict__ Y, double* __restrict__ sums) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n1 + n2) return;
  const double* z = i < n1 ? X + (size_t)i * L : Y + (size_t)(i - n1) * L;
  double s = 0.0;
  for (int k = 0; k < L; ++k) s += z[k];
  sums[i] = s != 0.0 ? s : 1.0;
}
__global__ __launch_bounds__(256) void k_mmd_rows(int n1, int n2, int L, const double* __restrict__ X,
                                                  const double* __restrict__ Y, double sigma,
                                                  const double* __restrict__ sums, double* __restrict__ part) {
  __shared__ double red[256];
  const int n = n1 + n2, i = blockIdx.x;
  const double* zi = i < n1 ? X + (size_t)i * L : Y + (size_t)(i - n1) * L;
  const double si = sums[i];
  double acc_x = 0.0, acc_y = 0.0;
  for (int j = 0; j < n; ++j) {
    const double* zj = j < n1 ? X + (size_t)j * L : Y + (size_t)(j - n1) * L;
    const double sj = sums[j];
    double d = 0.0;
    for (int k = threadIdx.x; k < L; k += 256) d += fabs(zi[k] / si - zj[k] / sj);
    __syncthreads();
    red[threadIdx.x] = d;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double dist = red[0] / 2.0;
      const double kv = exp(-dist * dist / (2.0 * sigma * sigma));
      if (j < n1) acc_x += kv; else acc_y += kv;
    }
  }
  if (threadIdx.x == 0) { part[2 * i] = acc_x; part[2 * i + 1] = acc_y; }
}
__global__ void k_mmd_finish(int n1, int n2, const double* __restrict__ part, double* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double xx = 0.0, xy = 0.0, yy = 0.0;
  for (int i = 0; i < n1; ++i) { xx += part[2 * i]; xy += part[2 * i + 1]; }
  for (int i = n1; i < n1 + n2; ++i) yy += part[2 * i + 1];
  out[0] = xx / ((double)n1 * n1) + yy / ((double)n2 * n2) - 2.0 * xy / ((double)n1 * n2);
}

}  // namespace

extern "C" {

size_t swg_graph_stats_workspace_bytes(int32_t B, int32_t N) {
  if (B <= 0 || N <= 0) return 0;
  return align256((size_t)B * N * ((N + 31) / 32) * sizeof(uint32_t));
}

int swg_graph_hists(int32_t B, int32_t N, int32_t cluster_bins, const float* adj, double* deg_hist, double* clu_hist,
                    double* lap, double iso_diag, void* ws, size_t ws_bytes, void* stream) {
  if (!adj || !deg_hist || !clu_hist || !ws) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || N > 32 * MAXW || cluster_bins <= 0 || cluster_bins > 4096) return SWG_E_SHAPE;
  if (ws_bytes < swg_graph_stats_workspace_bytes(B, N)) return SWG_E_WORKSPACE;
  const int W = (N + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* bits = (uint32_t*)ws;
  hipLaunchKernelGGL(k_graph_bits, dim3(min(cdiv(N * W, 256), 64), B), dim3(256), 0, s, N, W, adj, bits);
  const size_t lds = (size_t)(2 * N + cluster_bins + 1) * sizeof(int);
  hipLaunchKernelGGL(k_graph_deg_clu, dim3(B), dim3(256), lds, s, N, W, cluster_bins, bits, deg_hist, clu_hist, lap,
                     iso_diag);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_hist_pmf(int32_t B, int32_t n, int32_t bins, const double* vals, double first, double last, double snap,
                 double* out, void* stream) {
  if (!vals || !out) return SWG_E_NULL;
  if (B <= 0 || n <= 0 || bins <= 0 || bins > 8192 || !(last > first)) return SWG_E_SHAPE;
  hipLaunchKernelGGL(k_hist_pmf, dim3(B), dim3(256), (size_t)(bins + 1) * sizeof(int), (hipStream_t)stream, n, bins, vals,
                     first, last, snap, out);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_mmd_gaussian_tv(int32_t n1, int32_t n2, int32_t L, const double* X, const double* Y, double sigma, double* out,
                        double* scratch, void* stream) {
  if (!X || !Y || !out || !scratch) return SWG_E_NULL;
  if (n1 <= 0 || n2 <= 0 || L <= 0 || !(sigma > 0.0)) return SWG_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int n = n1 + n2;  // scratch: [n] row sums, then [2n] per-row partial kernel sums
  hipLaunchKernelGGL(k_mmd_row_sums, dim3(cdiv(n, 64)), dim3(64), 0, s, n1, n2, L, X, Y, scratch);
  hipLaunchKernelGGL(k_mmd_rows, dim3(n), dim3(256), 0, s, n1, n2, L, X, Y, sigma, scratch, scratch + n);
  hipLaunchKernelGGL(k_mmd_finish, dim3(1), dim3(64), 0, s, n1, n2, scratch + n, out);
  SWG_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
