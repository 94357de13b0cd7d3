// Time-lag correlations over sliding windows: the ensemble loops of the
// Green-Kubo (autocorrelation) and Einstein (mean-square displacement)
// calculators.
//
// A window w of a loaded batch covers frames w*ct .. w*ct + D - 1 (ct = correlation
// time, D = data range = tau[n_tau-1] + 1).  A block owns a few consecutive windows
// of a chunk of series, stages the frames they span in LDS as float64 once per
// series, and every thread evaluates lags against that image.
//
//   acf_partial_kernel   C[chunk][w][m] = sum_{s in chunk} sum_d sum_n y[n] y[n+m]
//                        (thread = (window, lag pair {m, R-1-m}): equal work per thread)
//   acf_finish_kernel    per window: c = scale/(R-m) * sum_chunks C, running sum of c
//                        (the calculator's acf_array), trapezoid running integral by
//                        a block scan, Welford mean / M2 of the integral over windows
//   acf_slide_kernel     identity lags and a small correlation time: consecutive windows
//                        overlap in all but 2*ct lag products, so a block computes its
//                        first window directly and then slides, fused with the
//                        per-window statistics of acf_finish_kernel
//   acf_band_kernel      many series: Q[t][m] = sum_s sum_d x_s[t] x_s[t+m], the lag
//                        products summed over series once, which acf_slide_kernel then
//                        consumes instead of the series themselves
//   msd_kernel           per block partial sums of squared displacements
//   combine kernels      fixed-order sums of the per-block partials
//
// All accumulation is float64 (fused multiply-add in the product loops); float32
// input is widened exactly.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "mdx_common.h"

using namespace mdx;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxOwned = 16;  // (window, lag pair) items per thread in the ACF kernel
constexpr int kSlidePerMax = 8;  // most lags a thread of the sliding kernel can own

// ------------------------------------------------------------------ staging

// Copies frames [frame0, frame0 + span) of one series into LDS as float64.
template <typename T>
__device__ __forceinline__ void stage_span(const T *__restrict__ series, int64_t frame0,
                                           int span, int k, double *__restrict__ dst) {
  const T *src = series + frame0 * k;
  const int n = span * k;
  for (int e = threadIdx.x; e < n; e += kThreads) dst[e] = static_cast<double>(src[e]);
}

// ---------------------------------------------------------------- ACF, stage A

template <typename T, int K>
__global__ __launch_bounds__(kThreads) void acf_partial_kernel(
    const T *__restrict__ x, int64_t n_series, int64_t n_frames, int k_runtime, int64_t ct,
    int64_t win_begin, int64_t n_win, int wpb, const int32_t *__restrict__ tau, int R,
    int D, int identity_tau, int64_t series_per_chunk, double *__restrict__ C) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int k = (K > 0) ? K : k_runtime;
  const int64_t wl0 = static_cast<int64_t>(blockIdx.x) * wpb;
  const int nw = static_cast<int>((n_win - wl0) < wpb ? (n_win - wl0) : wpb);
  const int64_t frame0 = (win_begin + wl0) * ct;
  const int span = static_cast<int>((nw - 1) * ct) + D;
  double *img = reinterpret_cast<double *>(lds_raw);
  int32_t *tau_l = reinterpret_cast<int32_t *>(img + static_cast<size_t>(((wpb - 1) * ct + D)) * k);
  for (int i = threadIdx.x; i < R; i += kThreads) tau_l[i] = tau[i];

  const int n_pairs = (R + 1) / 2;
  const int n_items = nw * n_pairs;
  double acc1[kMaxOwned], acc2[kMaxOwned];
#pragma unroll
  for (int o = 0; o < kMaxOwned; ++o) {
    acc1[o] = 0.0;
    acc2[o] = 0.0;
  }
  const int64_t s0 = static_cast<int64_t>(blockIdx.y) * series_per_chunk;
  const int64_t s1 = min(n_series, s0 + series_per_chunk);
  for (int64_t s = s0; s < s1; ++s) {
    __syncthreads();
    stage_span(x + s * n_frames * k, frame0, span, k, img);
    __syncthreads();
#pragma unroll
    for (int o = 0; o < kMaxOwned; ++o) {
      const int e = threadIdx.x + o * kThreads;
      if (e < n_items) {
        const int wl = e / n_pairs;
        const int p = e - wl * n_pairs;
        const int m1 = p;
        const int m2 = R - 1 - p;
        const double *base = img + static_cast<size_t>(wl) * ct * k;
        double sum = 0.0;
        if (identity_tau) {
          for (int n = 0; n < R - m1; ++n) {
            const double *a = base + n * k;
            const double *b = base + (n + m1) * k;
            for (int d = 0; d < k; ++d) sum = fma(a[d], b[d], sum);
          }
          acc1[o] += sum;
          if (m2 != m1) {
            sum = 0.0;
            for (int n = 0; n < R - m2; ++n) {
              const double *a = base + n * k;
              const double *b = base + (n + m2) * k;
              for (int d = 0; d < k; ++d) sum = fma(a[d], b[d], sum);
            }
            acc2[o] += sum;
          }
        } else {
          for (int n = 0; n < R - m1; ++n) {
            const double *a = base + tau_l[n] * k;
            const double *b = base + tau_l[n + m1] * k;
            for (int d = 0; d < k; ++d) sum = fma(a[d], b[d], sum);
          }
          acc1[o] += sum;
          if (m2 != m1) {
            sum = 0.0;
            for (int n = 0; n < R - m2; ++n) {
              const double *a = base + tau_l[n] * k;
              const double *b = base + tau_l[n + m2] * k;
              for (int d = 0; d < k; ++d) sum = fma(a[d], b[d], sum);
            }
            acc2[o] += sum;
          }
        }
      }
    }
  }
  double *Cb = C + (static_cast<int64_t>(blockIdx.y) * n_win + wl0) * R;
#pragma unroll
  for (int o = 0; o < kMaxOwned; ++o) {
    const int e = threadIdx.x + o * kThreads;
    if (e < n_items) {
      const int wl = e / n_pairs;
      const int p = e - wl * n_pairs;
      Cb[static_cast<int64_t>(wl) * R + p] = acc1[o];
      if (R - 1 - p != p) Cb[static_cast<int64_t>(wl) * R + (R - 1 - p)] = acc2[o];
    }
  }
}

// Exclusive prefix sum of one double per thread over the block, in thread order.
// `wave_tot` holds one slot per wave; the caller guarantees a barrier between two
// scans (the window loops have one).  Contains one barrier.
template <int kBlock>
__device__ __forceinline__ double block_exclusive_scan(double v, double *wave_tot) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double up = __shfl_up(incl, off);
    if (lane >= off) incl = incl + up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wave; ++w) base = base + wave_tot[w];
  return base + (incl - v);
}

// ---------------------------------------------------------------- ACF, stage B

// One block walks over a contiguous range of windows.  Per-block results:
//   acf_blk[block][R], mean_blk[block][R-1], m2_blk[block][R-1].
// Thread ownership of an index never changes, so the read-modify-write of these
// global arrays needs no synchronisation.
__global__ __launch_bounds__(kThreads) void acf_finish_kernel(
    const double *__restrict__ C, int n_chunks, int64_t n_win, int64_t windows_per_block,
    int R, double scale, double integral_scale, const double *__restrict__ time,
    double *__restrict__ acf_blk, double *__restrict__ mean_blk, double *__restrict__ m2_blk) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *cs = reinterpret_cast<double *>(lds_raw);  // [R] window correlation
  double *tot = cs + R;                              // [kThreads] segment totals
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * windows_per_block;
  const int64_t w1 = min(n_win, w0 + windows_per_block);
  double *acf = acf_blk + static_cast<int64_t>(blockIdx.x) * R;
  double *mean = mean_blk + static_cast<int64_t>(blockIdx.x) * (R - 1);
  double *m2 = m2_blk + static_cast<int64_t>(blockIdx.x) * (R - 1);
  for (int m = threadIdx.x; m < R; m += kThreads) acf[m] = 0.0;
  const int n_int = R - 1;
  const int seg = (n_int + kThreads - 1) / kThreads;
  const int i0 = min(n_int, static_cast<int>(threadIdx.x) * seg);
  const int i1 = min(n_int, i0 + seg);
  for (int i = i0; i < i1; ++i) {
    mean[i] = 0.0;
    m2[i] = 0.0;
  }
  for (int64_t w = w0; w < w1; ++w) {
    __syncthreads();
    for (int m = threadIdx.x; m < R; m += kThreads) {
      double c = 0.0;
      for (int ch = 0; ch < n_chunks; ++ch)
        c = c + C[(static_cast<int64_t>(ch) * n_win + w) * R + m];
      c = c * (scale / static_cast<double>(R - m));
      acf[m] = acf[m] + c;
      cs[m] = c * integral_scale;
    }
    if (time == nullptr) continue;  // uniform: no integral requested
    __syncthreads();
    double run = 0.0;
    for (int i = i0; i < i1; ++i)
      run = run + (time[i + 1] - time[i]) * (cs[i + 1] + cs[i]) / 2.0;
    run = block_exclusive_scan<kThreads>(run, tot);
    const double inv_count = 1.0 / static_cast<double>(w - w0 + 1);
    for (int i = i0; i < i1; ++i) {
      run = run + (time[i + 1] - time[i]) * (cs[i + 1] + cs[i]) / 2.0;
      const double delta = run - mean[i];
      const double mu = mean[i] + delta * inv_count;
      mean[i] = mu;
      m2[i] = m2[i] + delta * (run - mu);
    }
  }
}

// Merges the per-block results of acf_finish_kernel / acf_slide_kernel: out[0..R) =
// sum of the blocks' acf, out[R..2R-1) and out[2R-1..3R-2) = mean and M2 of the
// running integral over all windows (pairwise update of Chan et al.).
// One wave per lag / integral index: lane l folds blocks l, l + 64, ... in order,
// then the 64 partial triples are combined by a fixed butterfly.
__global__ __launch_bounds__(kThreads) void acf_merge_kernel(
    const double *__restrict__ acf_blk, const double *__restrict__ mean_blk,
    const double *__restrict__ m2_blk, const int64_t *__restrict__ counts, int64_t n_blocks,
    int R, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (i >= R) return;
  const bool moments = (mean_blk != nullptr) && (i < R - 1);
  double total = 0.0;
  double n_a = 0.0, mean_a = 0.0, m2_a = 0.0;
  for (int64_t b = lane; b < n_blocks; b += 64) {
    total = total + acf_blk[b * R + i];
    if (!moments) continue;
    const double n_b = static_cast<double>(counts[b]);
    if (n_b <= 0.0) continue;
    const double mean_b = mean_blk[b * (R - 1) + i];
    const double m2_b = m2_blk[b * (R - 1) + i];
    const double delta = mean_b - mean_a;
    const double n_ab = n_a + n_b;
    mean_a = mean_a + delta * (n_b / n_ab);
    m2_a = m2_a + m2_b + delta * delta * (n_a * n_b / n_ab);
    n_a = n_ab;
  }
  for (int off = 32; off > 0; off >>= 1) {
    total = total + __shfl_xor(total, off);
    const double n_b = __shfl_xor(n_a, off);
    const double mean_b = __shfl_xor(mean_a, off);
    const double m2_b = __shfl_xor(m2_a, off);
    const double n_ab = n_a + n_b;
    if (n_ab > 0.0) {
      // symmetric in (a, b): both partners arrive at the same triple
      const double mean_ab = (n_a * mean_a + n_b * mean_b) / n_ab;
      const double delta = mean_b - mean_a;
      m2_a = m2_a + m2_b + delta * delta * (n_a * n_b / n_ab);
      mean_a = mean_ab;
      n_a = n_ab;
    }
  }
  if (lane == 0) {
    out[i] = total;
    if (moments) {
      out[R + i] = mean_a;
      out[R + (R - 1) + i] = m2_a;
    }
  }
}

// ------------------------------------------------------- ACF, sliding windows

// Lag product P(t, m) = sum_s sum_d x_s[t][d] * x_s[t + m][d], from the series.
// K > 0: the number of components at compile time; kSingle: exactly one series
// (both let the compiler unroll the callers' loops over consecutive frames).
template <typename T, int K, bool kSingle>
struct SeriesProduct {
  const T *x;
  int64_t n_series, n_frames;
  int k_runtime;
  __device__ __forceinline__ double one(const T *a, int m) const {
    const int k = (K > 0) ? K : k_runtime;
    const T *b = a + static_cast<int64_t>(m) * k;
    double sum = 0.0;
#pragma unroll
    for (int d = 0; d < k; ++d)
      sum = fma(static_cast<double>(a[d]), static_cast<double>(b[d]), sum);
    return sum;
  }
  __device__ __forceinline__ double operator()(int64_t t, int m) const {
    const int k = (K > 0) ? K : k_runtime;
    if (kSingle) return one(x + t * k, m);
    double sum = 0.0;
    for (int64_t s = 0; s < n_series; ++s) sum = sum + one(x + (s * n_frames + t) * k, m);
    return sum;
  }
};

// The same product read from the band table of acf_band_kernel (frames t0 ..
// t0 + t_len - 1), after its series-chunk planes have been summed.
struct BandProduct {
  const double *Q;
  int64_t t0;
  int R;
  __device__ __forceinline__ double operator()(int64_t t, int m) const {
    return Q[(t - t0) * R + m];
  }
};

// Cross product of two series minus a weighted band table:
//   P(t, m) = cross_weight * sum_d a[t + m][d] * b[t][d] - self_weight * Q[t - t0][m]
// (Q may be null).  a, b are [n_frames][k].
template <typename T>
struct DistinctProduct {
  const T *a, *b;
  int k;
  double cross_weight;
  const double *Q;
  int64_t t0;
  int R;
  double self_weight;
  __device__ __forceinline__ double operator()(int64_t t, int m) const {
    const T *pa = a + (t + m) * k;
    const T *pb = b + t * k;
    double sum = 0.0;
    for (int d = 0; d < k; ++d)
      sum = fma(static_cast<double>(pa[d]), static_cast<double>(pb[d]), sum);
    sum = sum * cross_weight;
    if (Q != nullptr) sum = sum - self_weight * Q[(t - t0) * R + m];
    return sum;
  }
};

// Q[0][e] = sum_chunks Q[chunk][e], planes in index order.
__global__ __launch_bounds__(kThreads) void band_combine_kernel(double *__restrict__ Q,
                                                                int n_chunks, int64_t plane) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= plane) return;
  double sum = Q[e];
  for (int ch = 1; ch < n_chunks; ++ch) sum = sum + Q[static_cast<int64_t>(ch) * plane + e];
  Q[e] = sum;
}

// lag_weight[m] = scale / (R - m) (tfp's unbiased estimate) or scale.
__global__ __launch_bounds__(kThreads) void lag_weight_kernel(double *__restrict__ lag_weight,
                                                              int R, double scale,
                                                              int lag_normalise) {
  const int m = blockIdx.x * kThreads + threadIdx.x;
  if (m < R) lag_weight[m] = lag_normalise ? scale / static_cast<double>(R - m) : scale;
}

// One block walks over windows [w0, w1) of the launch.  W[m] = sum_{n < R - m}
// P(o + n, m) for the window origin o is computed directly for the first window
// and updated by the 2*ct products that leave and enter for every later one.
// Outputs as acf_finish_kernel, at block index block_offset + blockIdx.x.
template <typename Product, int kBlock, int kPer>
__global__ __launch_bounds__(kBlock) void acf_slide_kernel(
    Product prod, int64_t ct, int64_t win_begin, int64_t n_win, int64_t windows_per_block,
    int R, const double *__restrict__ lag_weight, double integral_scale,
    const double *__restrict__ time, int64_t block_offset, double *__restrict__ acf_blk,
    double *__restrict__ mean_blk, double *__restrict__ m2_blk) {
  // A thread owns the lags m = tid + j * kBlock and the integral indices
  // i0 + j, j < kPer.  Four running values per lag stay in registers (window sum,
  // its total over windows, mean and M2 of the integral): with 8 lags that is half of
  // the 128 registers a lane of a 1024-thread block has, which is why the per-lag
  // constants (weight, time step) are read from their tables in every window rather
  // than kept.  LDS holds two images of the window correlation (so one barrier per
  // window separates writers from readers) and the wave totals of the scan.
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *cs_base = reinterpret_cast<double *>(lds_raw);  // [2][R]
  double *tot = cs_base + 2 * static_cast<size_t>(R);     // [kBlock / 64]
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * windows_per_block;
  const int64_t w1 = min(n_win, w0 + windows_per_block);
  const int64_t blk = block_offset + blockIdx.x;
  const int tid = threadIdx.x;
  const int n_int = R - 1;
  const int seg = (n_int + kBlock - 1) / kBlock;  // <= kPer, chosen by the host
  const int i0 = min(n_int, tid * seg);
  const int i1 = min(n_int, i0 + seg);

  double W[kPer], acf[kPer], mean[kPer], m2[kPer];
  const int64_t o0 = (win_begin + w0) * ct;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int m = tid + j * kBlock;
    acf[j] = 0.0;
    mean[j] = 0.0;
    m2[j] = 0.0;
    W[j] = 0.0;
    if (m < R) {
      double sum = 0.0;
      for (int n = 0; n < R - m; ++n) sum = sum + prod(o0 + n, m);
      W[j] = sum;
    }
  }
  for (int64_t w = w0; w < w1; ++w) {
    const int64_t o_prev = (win_begin + w - 1) * ct;
    double *cs = cs_base + static_cast<size_t>((w - w0) & 1) * R;
    // an offset the compiler cannot see through keeps the table reads inside the loop
    int opaque = 0;
    asm volatile("" : "+s"(opaque));
    const double *weight = lag_weight + opaque;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int m = tid + j * kBlock;
      if (m < R) {
        if (w > w0) {
          double d = 0.0;
          for (int64_t n = 0; n < ct; ++n)
            d = d + (prod(o_prev + (R - m) + n, m) - prod(o_prev + n, m));
          W[j] = W[j] + d;
        }
        const double c = W[j] * weight[m];
        acf[j] = acf[j] + c;
        cs[m] = c * integral_scale;
      }
    }
    if (time == nullptr) continue;  // uniform: no integral requested
    const double *tm = time + opaque;
    __syncthreads();  // cs complete; the other image was last read two barriers ago
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (i0 + j < i1)
        run = run + (tm[i0 + j + 1] - tm[i0 + j]) * (cs[i0 + j + 1] + cs[i0 + j]) / 2.0;
    }
    run = block_exclusive_scan<kBlock>(run, tot);
    const double inv_count = 1.0 / static_cast<double>(w - w0 + 1);
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (i0 + j < i1) {
        run = run + (tm[i0 + j + 1] - tm[i0 + j]) * (cs[i0 + j + 1] + cs[i0 + j]) / 2.0;
        const double delta = run - mean[j];
        const double mu = mean[j] + delta * inv_count;
        mean[j] = mu;
        m2[j] = m2[j] + delta * (run - mu);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int m = tid + j * kBlock;
    if (m < R) acf_blk[blk * R + m] = acf[j];
    if (i0 + j < i1) {
      mean_blk[blk * (R - 1) + i0 + j] = mean[j];
      m2_blk[blk * (R - 1) + i0 + j] = m2[j];
    }
  }
}

// ------------------------------------------- ACF, sliding windows in strips
//
// The same recurrence for one series, organised so that the per-window prefix sum
// of the integral does not cost a block-wide scan per window.  A block walks over its
// windows in strips of kStripWin; within a strip the lags are taken in phases of
// 64 * kWaves, one 16 x 64 tile (windows x lags) per wave:
//   pass 1  lane = lag: the products that enter and leave, W, the window correlation
//           c = W * weight, its total over windows; c goes to the wave's LDS tile;
//   pass 2  lane = (window, quarter of the tile's lags): sequential prefix of the
//           trapezoid terms along the lags, J_m = sum_{j<m} c_j g_j + c_m h_m, which is
//           the cumulative integral I_{m-1}; quarters are joined by three shuffles,
//           tiles by the row totals every wave posts in LDS (one barrier per phase);
//   pass 3  lane = lag: sums of (J - ref) and (J - ref)^2 over the strip's windows,
//           ref = J of the block's first window (keeps the sums small).
// Per-lag running values live in the block's output rows and two state rows (L2
// traffic of ten values per lag per 16 windows).  The series is read as float64
// component planes xs[d * pitch + t - t0].

constexpr int kStripWin = 16;
constexpr int kTilePitch = 65;  // doubles per tile row: rows land on different banks

__host__ __device__ constexpr size_t strip_lds_bytes(int waves) {
  return sizeof(double) * (static_cast<size_t>(waves) * kStripWin * kTilePitch +
                           2 * static_cast<size_t>(waves) * kStripWin);
}

// Component planes of frames [t0, t0 + t_len): planes[d * pitch + kPlaneFront + i] =
// x[(t0 + i) * k + d], in the input's precision.  The pads before and after a plane (zeroed by
// the host) let the blocked loads of acf_strip_kernel run past the ends.
constexpr int kPlaneFront = 2;
constexpr int kPlaneBack = 126;

template <typename T>
__global__ __launch_bounds__(kThreads) void series_planes_kernel(const T *__restrict__ x, int k,
                                                                 int64_t t0, int64_t t_len,
                                                                 int64_t pitch,
                                                                 T *__restrict__ planes) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= t_len * k) return;
  const int64_t i = e / k;
  const int d = static_cast<int>(e - i * k);
  planes[d * pitch + kPlaneFront + i] = x[(t0 + i) * k + d];
}

// Trapezoid coefficients over the lag times: g[m] = (dt[m-1] + dt[m]) / 2 with the
// missing neighbour taken as 0, h[m] = dt[m-1] / 2; zero from R up to `padded`.
__global__ __launch_bounds__(kThreads) void trapezoid_coeff_kernel(
    const double *__restrict__ time, int R, int padded, double *__restrict__ g,
    double *__restrict__ h) {
  const int m = blockIdx.x * kThreads + threadIdx.x;
  if (m >= padded) return;
  double gv = 0.0, hv = 0.0;
  if (m < R) {
    const double before = m >= 1 ? time[m] - time[m - 1] : 0.0;
    const double after = m + 1 < R ? time[m + 1] - time[m] : 0.0;
    gv = (before + after) / 2.0;
    hv = before / 2.0;
  }
  g[m] = gv;
  h[m] = hv;
}

template <typename TP, int K, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void acf_strip_kernel(
    const TP *__restrict__ xs_padded, int64_t pitch, int k_runtime, int64_t t0, int64_t ct,
    int64_t win_begin, int64_t n_win, int64_t windows_per_block, int R,
    const double *__restrict__ weight, const double *__restrict__ gcoef,
    const double *__restrict__ hcoef, double integral_scale, int64_t block_offset,
    double *__restrict__ acf_blk, double *__restrict__ mean_blk, double *__restrict__ m2_blk,
    double *__restrict__ w_state, double *__restrict__ ref_state) {
  constexpr int kLagsPerPhase = kWaves * 64;
  const TP *__restrict__ xs = xs_padded + kPlaneFront;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *tile = reinterpret_cast<double *>(lds_raw) +
                 static_cast<size_t>(threadIdx.x >> 6) * kStripWin * kTilePitch;
  double *rowtot = reinterpret_cast<double *>(lds_raw) +
                   static_cast<size_t>(kWaves) * kStripWin * kTilePitch;  // [2][kWaves][16]
  const int k = (K > 0) ? K : k_runtime;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t w0 = static_cast<int64_t>(blockIdx.x) * windows_per_block;
  const int64_t w1 = min(n_win, w0 + windows_per_block);
  const int64_t blk = block_offset + blockIdx.x;
  const bool integrate = gcoef != nullptr;
  const int n_phases = (R + kLagsPerPhase - 1) / kLagsPerPhase;
  double *acf_b = acf_blk + blk * R;
  double *mean_b = mean_blk + blk * (R - 1) - 1;  // indexed by lag m >= 1
  double *m2_b = m2_blk + blk * (R - 1) - 1;
  double *w_b = w_state + static_cast<int64_t>(blockIdx.x) * R;  // state rows: per launch
  double *ref_b = ref_state + static_cast<int64_t>(blockIdx.x) * R;

  // first window of the block, directly: eight products at a time per lag; the
  // wave runs to the trip count of its first lane and masks the rest
  {
    const TP *a = xs + ((win_begin + w0) * ct - t0);
    for (int m = tid; m < R; m += kLagsPerPhase) {
      const int trips = R - (m - lane);  // of lane 0 of this wave
      const int mine = R - m;
      const int common = max(0, trips - 63) & ~7;  // every lane of the wave runs this far
      double sum = 0.0;
      for (int n0 = 0; n0 < trips; n0 += 8) {
        double p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = 0.0;
#pragma unroll
        for (int d = 0; d < k; ++d) {
          const TP *u = a + d * pitch + n0;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            p[i] = fma(static_cast<double>(u[i]), static_cast<double>(u[i + m]), p[i]);
        }
        if (n0 < common) {
#pragma unroll
          for (int i = 0; i < 8; ++i) sum = sum + p[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (n0 + i < mine) sum = sum + p[i];
        }
      }
      w_b[m] = sum;
      acf_b[m] = 0.0;
      if (integrate && m >= 1) {
        mean_b[m] = 0.0;
        m2_b[m] = 0.0;
      }
    }
  }

  const int w_i = lane & (kStripWin - 1);  // pass 2: this lane's window of the strip
  const int quarter = lane >> 4;           // and its 16 lags of the tile
  int parity = 0;
  for (int64_t ws = w0; ws < w1; ws += kStripWin) {
    const int rows = static_cast<int>(min<int64_t>(kStripWin, w1 - ws));
    double carry = 0.0;  // integral terms of window ws + w_i over the lags of earlier phases
    for (int p = 0; p < n_phases; ++p) {
      const int m_base = p * kLagsPerPhase + wave * 64;
      const bool tile_ok = m_base < R;  // wave-uniform
      const int m = m_base + lane;
      const bool lag_ok = m < R;
      if (tile_ok) {
        double W = 0.0, acf = 0.0, wt = 0.0;
        if (lag_ok) {
          W = w_b[m];
          acf = acf_b[m];
          wt = weight[m];
        }
        if (ct == 1) {
          // origins advance by one frame per window: a lag needs kSub consecutive
          // values of each operand for kSub windows (the operands shared by all lags
          // sit in scalar registers, which bounds kSub)
          constexpr int kSub = 4;
          const TP *leave0 = xs + (win_begin + ws - 1 - t0);
          const int ma = lag_ok ? m : 0;  // lanes past the last lag read valid addresses
          for (int sub = 0; sub < rows; sub += kSub) {
            double in[kSub], out[kSub];
#pragma unroll
            for (int i = 0; i < kSub; ++i) {
              in[i] = 0.0;
              out[i] = 0.0;
            }
            const TP *leave = leave0 + sub;
            const TP *enter = leave + R;
#pragma unroll
            for (int d = 0; d < k; ++d) {
              const TP *lv = leave + d * pitch;
              const TP *en = enter + d * pitch;
#pragma unroll
              for (int i = 0; i < kSub; ++i) {
                in[i] = fma(static_cast<double>(en[i - ma]), static_cast<double>(en[i]), in[i]);
                out[i] = fma(static_cast<double>(lv[i]), static_cast<double>(lv[i + ma]), out[i]);
              }
            }
            // lanes past the last lag carry a meaningless W; their weight is zero
            if (sub + kSub <= rows && (ws > w0 || sub > 0)) {
#pragma unroll
              for (int i = 0; i < kSub; ++i) {
                W = W + (in[i] - out[i]);
                const double c = W * wt;
                acf = acf + c;
                if (integrate) tile[(sub + i) * kTilePitch + lane] = c * integral_scale;
              }
            } else {
#pragma unroll
              for (int i = 0; i < kSub; ++i) {
                const int row = sub + i;
                if (row < rows) {
                  if (ws + row > w0) W = W + (in[i] - out[i]);
                  const double c = W * wt;
                  acf = acf + c;
                  if (integrate) tile[row * kTilePitch + lane] = c * integral_scale;
                }
              }
            }
          }
        } else {
          for (int i = 0; i < rows; ++i) {
            const int64_t w = ws + i;
            if (w > w0 && lag_ok) {
              const TP *leave = xs + ((win_begin + w - 1) * ct - t0);
              const TP *enter = leave + R;
              double delta = 0.0;
              for (int64_t n = 0; n < ct; ++n) {
                double in = 0.0, out = 0.0;
#pragma unroll
                for (int d = 0; d < k; ++d) {
                  in = fma(static_cast<double>(enter[d * pitch + n - m]),
                           static_cast<double>(enter[d * pitch + n]), in);
                  out = fma(static_cast<double>(leave[d * pitch + n]),
                            static_cast<double>(leave[d * pitch + n + m]), out);
                }
                delta = delta + (in - out);
              }
              W = W + delta;
            }
            const double c = W * wt;
            acf = acf + c;
            if (integrate) tile[i * kTilePitch + lane] = c * integral_scale;
          }
        }
        if (lag_ok) {
          w_b[m] = W;
          acf_b[m] = acf;
        }
      }
      if (!integrate) continue;  // uniform

      // pass 2: the tile is private to the wave, whose LDS accesses execute in order
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double J[16];
      double run = 0.0;
      if (tile_ok) {
        const double *row = tile + w_i * kTilePitch + quarter * 16;
        const double *g = gcoef + m_base + quarter * 16;
        const double *h = hcoef + m_base + quarter * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const double c = row[j];
          J[j] = fma(c, h[j], run);
          run = fma(c, g[j], run);
        }
      }
      const double q1 = __shfl_up(run, 16);
      const double q2 = __shfl_up(run, 32);
      const double q3 = __shfl_up(run, 48);
      double offset = 0.0;
      if (quarter >= 3) offset = offset + q3;
      if (quarter >= 2) offset = offset + q2;
      if (quarter >= 1) offset = offset + q1;
      double *posted = rowtot + static_cast<size_t>(parity) * kWaves * kStripWin;
      if (quarter == 3) posted[wave * kStripWin + w_i] = tile_ok ? offset + run : 0.0;
      __syncthreads();
      double before = carry;
      double total = carry;
#pragma unroll
      for (int v = 0; v < kWaves; ++v) {
        if (v == wave) before = total;
        total = total + posted[v * kStripWin + w_i];
      }
      carry = total;
      parity ^= 1;
      if (tile_ok) {
        double *row = tile + w_i * kTilePitch + quarter * 16;
        const double base = before + offset;
#pragma unroll
        for (int j = 0; j < 16; ++j) row[j] = J[j] + base;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      // pass 3
      if (lag_ok && m >= 1) {
        double s1 = mean_b[m], s2 = m2_b[m];
        double ref = (ws == w0) ? tile[lane] : ref_b[m];
        for (int i = 0; i < rows; ++i) {
          const double d = tile[i * kTilePitch + lane] - ref;
          s1 = s1 + d;
          s2 = fma(d, d, s2);
        }
        mean_b[m] = s1;
        m2_b[m] = s2;
        if (ws == w0) ref_b[m] = ref;
      }
    }
  }

  if (integrate) {
    const double count = static_cast<double>(w1 - w0);
    for (int m = tid; m < R; m += kLagsPerPhase) {
      if (m < 1) continue;
      const double s1 = mean_b[m], s2 = m2_b[m];
      const double spread = s2 - s1 * s1 / count;
      mean_b[m] = ref_b[m] + s1 / count;
      m2_b[m] = spread > 0.0 ? spread : 0.0;
    }
  }
}

// Band kernel.  For frames t (relative to t0) and lags m, summed over a chunk of series:
//   kMsd == false   Q[chunk][t][m] = sum_s sum_d x_s[t][d] * x_s[t + m][d]     (t < t_len, m < R)
//   kMsd == true    partial[chunk * tiles_t + tile][m] = sum_{t in tile, t < t_len}
//                                           sum_s sum_d (x_s[t + m][d] - x_s[t][d])^2
// A block owns a tile of 8*TG frames x 8*MG lags (TG * MG = 256 threads) and a chunk of
// series; a thread owns an 8 x 8 patch of it, whose 64 terms per dimension need only
// 8 + 15 staged values because x_s[t + m] depends on t + m.  The frames of one series
// are staged per dimension with one pad slot after every 8 frames, which spreads the
// 8-frame lane stride over all LDS banks.
// K = number of components when known at compile time (the next series is then
// fetched into registers while the current one is being processed), 0 otherwise.
// Block index -> (frame tile, lag tile, series chunk) of the band kernels.  The grid is
// one-dimensional and a multiple of eight.  Blocks with equal blockIdx.x % 8 share an XCD
// and its L2: each such group takes a contiguous eighth of the items in the order
// (chunk, frame tile, lag tile), so the blocks running on an XCD at any time stage
// neighbouring frame tiles of the same series and find most of their frames in that L2
// (a frame is staged by up to six frame tiles and every lag tile).  Returns false for the
// padding blocks of the last eighth.
__device__ __forceinline__ bool band_item(int64_t tiles_t, int tiles_m, int n_chunks,
                                          int64_t *tile_t_index, int *tile_m_index, int *chunk) {
  const int64_t n_items = tiles_t * tiles_m * n_chunks;
  const int64_t per_group = (n_items + 7) / 8;
  const int64_t within = blockIdx.x / 8;
  const int64_t item = static_cast<int64_t>(blockIdx.x % 8) * per_group + within;
  if (within >= per_group || item >= n_items) return false;
  *tile_m_index = static_cast<int>(item % tiles_m);
  const int64_t rest = item / tiles_m;
  *tile_t_index = rest % tiles_t;
  *chunk = static_cast<int>(rest / tiles_t);
  return true;
}

inline unsigned band_grid(int64_t tiles_t, int64_t tiles_m, int64_t n_chunks) {
  return static_cast<unsigned>(8 * ceil_div(tiles_t * tiles_m * n_chunks, 8));
}

template <int MG>
struct BandTile {
  static constexpr int TG = kThreads / MG;
  static constexpr int T = 8 * TG;      // frames per tile
  static constexpr int M = 8 * MG;      // lags per tile
  static constexpr int kStage = T + (T + M);  // staged frames: x[t], then x[t + m]
  static constexpr int kPitch = kStage + kStage / 8;
  static size_t lds_bytes(int k) {
    const size_t image = sizeof(double) * 2 * k * kPitch;
    const size_t reduce = sizeof(double) * TG * M;
    return image > reduce ? image : reduce;
  }
};

template <typename T, int K, int MG, bool kMsd>
__global__ __launch_bounds__(kThreads) void acf_band_kernel(
    const T *__restrict__ x, int64_t n_series, int64_t n_frames, int k_runtime, int64_t t0,
    int64_t t_len, int R, int64_t series_per_chunk, double *__restrict__ Q, int64_t tiles_t,
    int tiles_m, int n_chunks) {
  using Tile = BandTile<MG>;
  int64_t block_t;
  int block_m, block_chunk;
  if (!band_item(tiles_t, tiles_m, n_chunks, &block_t, &block_m, &block_chunk)) return;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *img = reinterpret_cast<double *>(lds_raw);  // [2][k][pitch]
  const int k = (K > 0) ? K : k_runtime;
  const int mg = threadIdx.x % MG;
  const int tg = threadIdx.x / MG;
  const int64_t tile_t = block_t * Tile::T;
  const int tile_m = block_m * Tile::M;
  const int64_t s0 = static_cast<int64_t>(block_chunk) * series_per_chunk;
  const int64_t s1 = min(n_series, s0 + series_per_chunk);
  // staged per series: the tile's frames x[t] first, then the T + M frames x[t + m]
  // starting at frame0
  const int64_t frame0 = t0 + tile_t + tile_m;
  constexpr int n_stage = Tile::kStage;
  constexpr int pitch = Tile::kPitch;
  constexpr int kOwn = (n_stage + kThreads - 1) / kThreads;  // frames a thread stages
  const int plane = k * pitch;

  // the frames this thread stages (the same for every series)
  int64_t own_frame[kOwn];
  int own_slot[kOwn];
#pragma unroll
  for (int o = 0; o < kOwn; ++o) {
    const int f = threadIdx.x + o * kThreads;
    const int64_t frame = (f < Tile::T) ? (t0 + tile_t + f) : (frame0 + (f - Tile::T));
    own_frame[o] = (f < n_stage && frame < n_frames) ? frame : -1;
    own_slot[o] = (f < n_stage) ? f + (f >> 3) : -1;
  }

  double acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;

  T pre[kOwn][K > 0 ? K : 1];
  auto fetch = [&](int64_t s) {
    const T *series = x + s * n_frames * k;
#pragma unroll
    for (int o = 0; o < kOwn; ++o)
#pragma unroll
      for (int d = 0; d < K; ++d)
        pre[o][d] = (own_frame[o] >= 0) ? series[own_frame[o] * K + d] : T(0);
  };
  if (K > 0 && s0 < s1) fetch(s0);

  int buf = 0;
  for (int64_t s = s0; s < s1; ++s) {
    double *dst = img + static_cast<size_t>(buf) * plane;
    if (K > 0) {
#pragma unroll
      for (int o = 0; o < kOwn; ++o)
        if (own_slot[o] >= 0) {
#pragma unroll
          for (int d = 0; d < K; ++d) dst[d * pitch + own_slot[o]] = static_cast<double>(pre[o][d]);
        }
    } else {
      const T *series = x + s * n_frames * k;
#pragma unroll
      for (int o = 0; o < kOwn; ++o)
        if (own_slot[o] >= 0) {
          for (int d = 0; d < k; ++d)
            dst[d * pitch + own_slot[o]] =
                (own_frame[o] >= 0) ? static_cast<double>(series[own_frame[o] * k + d]) : 0.0;
        }
    }
    __syncthreads();
    if (K > 0 && s + 1 < s1) fetch(s + 1);
    for (int d = 0; d < k; ++d) {
      const double *row = dst + d * pitch;
      double a[8], b[15];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = row[9 * tg + i];
      const double *rb = row + (Tile::T + Tile::T / 8) + 9 * (tg + mg);
#pragma unroll
      for (int c = 0; c < 15; ++c) b[c] = rb[c + (c >> 3)];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (kMsd) {
            const double diff = b[i + j] - a[i];
            acc[i][j] = fma(diff, diff, acc[i][j]);
          } else {
            acc[i][j] = fma(a[i], b[i + j], acc[i][j]);
          }
        }
    }
    buf ^= 1;  // the other image was last read before this iteration's barrier
  }
  if (!kMsd) {
    double *out = Q + static_cast<int64_t>(block_chunk) * t_len * R;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t t = tile_t + 8 * tg + i;
      if (t >= t_len) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = tile_m + 8 * mg + j;
        if (m < R) out[t * R + m] = acc[i][j];
      }
    }
    return;
  }
  // sum over the tile's frames: first this thread's 8, then the TG thread groups
  __syncthreads();  // the staged images are no longer read
  double *red = img;  // [TG][M]
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (tile_t + 8 * tg + i < t_len) sum = sum + acc[i][j];
    red[tg * Tile::M + 8 * mg + j] = sum;
  }
  __syncthreads();
  double *out = Q + (static_cast<int64_t>(block_chunk) * tiles_t + block_t) * R;
  for (int m = threadIdx.x; m < Tile::M; m += kThreads) {
    if (tile_m + m >= R) continue;
    double sum = 0.0;
    for (int g = 0; g < Tile::TG; ++g) sum = sum + red[g * Tile::M + m];
    out[tile_m + m] = sum;
  }
}

// The band table of three-component series on the matrix cores.  With the rows
// (series, component) as the inner dimension, Q[t][t + m] is the Gram matrix X^T X
// restricted to a band, so 16 x 16 tiles of it are v_mfma_f64_16x16x4_f64 products of
// staged frame blocks: A = the tile's own frames (one block of 16 per wave), B = the
// frames 16 j further on.  A block owns 64 frames x 256 lags: wave w holds the 17
// 16 x 16 tiles (frames 16w.., frames 16(w + j).., j = 0..16) that cover its lags, the
// first and last of them half used.  Eight series (24 rows = 6 inner blocks of 4) are
// staged per round in the input precision, double buffered, the next round
// prefetched into registers.
// Operand layout of the instruction (checked on gfx950): lane l supplies
// A[l % 16][l / 16] and B[l / 16][l % 16] and receives D[l / 16 + 4 r][l % 16], r < 4.
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kMfT = 64;
constexpr int kMfM = 256;
constexpr int kMfTiles = kMfM / 16 + 1;          // 16 x 16 tiles per wave
constexpr int kMfSeries = 8;                     // series per staging round
constexpr int kMfRows = 3 * kMfSeries;
constexpr int kMfFrames = kMfT + (kMfT + kMfM);  // 64 own frames, then 320 shifted ones
constexpr int kMfPerLane = kMfFrames / 64;       // frames a thread stages per series
// row pitch: the four rows of an inner block land 16 (float) / 32 (double) banks apart
constexpr int kMfPitch = kMfFrames + 16;

template <typename T>
__global__ __launch_bounds__(kThreads) void acf_band_mfma_kernel(
    const T *__restrict__ x, int64_t n_series, int64_t n_frames, int64_t t0, int64_t t_len,
    int R, int64_t series_per_chunk, double *__restrict__ Q, int64_t tiles_t, int tiles_m,
    int n_chunks) {
  int64_t block_t;
  int block_m, block_chunk;
  if (!band_item(tiles_t, tiles_m, n_chunks, &block_t, &block_m, &block_chunk)) return;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // staged in the input type (converted when read): [2][kMfRows][kMfPitch]
  T(*img)[kMfRows][kMfPitch] = reinterpret_cast<T(*)[kMfRows][kMfPitch]>(lds_raw);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t tile_t = block_t * kMfT;
  const int tile_m = block_m * kMfM;
  const int64_t s0 = static_cast<int64_t>(block_chunk) * series_per_chunk;
  const int64_t s1 = min(n_series, s0 + series_per_chunk);

  // staging: thread = (series slots wave, wave + 4, frames lane + 64 q)
  constexpr int kSlots = kMfSeries / 4;
  int64_t own_frame[kMfPerLane];
#pragma unroll
  for (int q = 0; q < kMfPerLane; ++q) {
    const int f = lane + 64 * q;
    const int64_t frame = (f < kMfT) ? (t0 + tile_t + f) : (t0 + tile_t + tile_m + (f - kMfT));
    own_frame[q] = (frame < n_frames) ? frame : -1;
  }
  T pre[kSlots][kMfPerLane][3];
  auto fetch = [&](int64_t s_base) {
#pragma unroll
    for (int u = 0; u < kSlots; ++u) {
      const int64_t s = s_base + wave + 4 * u;
      const T *series = x + s * n_frames * 3;
#pragma unroll
      for (int q = 0; q < kMfPerLane; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          pre[u][q][d] = (s < s1 && own_frame[q] >= 0) ? series[own_frame[q] * 3 + d] : T(0);
    }
  };

  v4d acc[kMfTiles];
#pragma unroll
  for (int j = 0; j < kMfTiles; ++j) acc[j] = v4d{0.0, 0.0, 0.0, 0.0};

  if (s0 < s1) fetch(s0);
  int buf = 0;
  const int kk = lane >> 4;
  const int col = lane & 15;
  for (int64_t s_base = s0; s_base < s1; s_base += kMfSeries) {
#pragma unroll
    for (int u = 0; u < kSlots; ++u)
#pragma unroll
      for (int q = 0; q < kMfPerLane; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) img[buf][(wave + 4 * u) * 3 + d][lane + 64 * q] = pre[u][q][d];
    __syncthreads();
    if (s_base + kMfSeries < s1) fetch(s_base + kMfSeries);
#pragma unroll
    for (int g = 0; g < kMfRows / 4; ++g) {
      const T *row = img[buf][4 * g + kk];
      const double a = static_cast<double>(row[16 * wave + col]);
#pragma unroll
      for (int j = 0; j < kMfTiles; ++j) {
        const double b = static_cast<double>(row[kMfT + 16 * (wave + j) + col]);
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      }
    }
    buf ^= 1;  // the other image was last read before this round's barrier
  }
  double *out = Q + static_cast<int64_t>(block_chunk) * t_len * R;
#pragma unroll
  for (int j = 0; j < kMfTiles; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int trow = kk + 4 * r;
      const int m_rel = 16 * j + col - trow;
      const int64_t t = tile_t + 16 * wave + trow;
      const int m = tile_m + m_rel;
      if (m_rel >= 0 && m_rel < kMfM && m < R && t < t_len) out[t * R + m] = acc[j][r];
    }
  }
}

// Mean-square displacement of three-component series on the matrix cores:
//   partial[chunk * tiles_t + tile][m] = sum_{t in tile, t < t_len} sum_s |x_s[t + m] - x_s[t]|^2
// through |b - a|^2 = |a|^2 + |b|^2 - 2 a.b, the cross term being the band of the Gram matrix
// that acf_band_mfma_kernel forms (same tiles, staging and operand layout, lag tile kM).
// The expansion cancels, so every series is first referred to its position at the
// tile's first frame: the staged values are then displacements over at most
// 128 + kM frames, of the size of the result itself, and the three terms lose a few of
// float64's sixteen digits instead of the square of (position / displacement).  The
// reference of each staged row sits beside the image and is subtracted, in float64,
// when an operand is read; the squares |x - ref|^2 are summed per frame by the staging
// threads.  Lag 0 is written as the exact zero it is.
template <int kM>
struct MsdMf {
  static constexpr int kT = 64;
  static constexpr int kTiles = kM / 16 + 1;
  static constexpr int kFrames = kT + (kT + kM);
  static constexpr int kPerLane = (kFrames + 63) / 64;  // staging slots per lane and row
  static constexpr int kPitch = 64 * kPerLane + 16;      // rows 16 banks apart
  static constexpr size_t kRefBytes = sizeof(double) * 2 * kMfRows;
  static size_t lds_bytes(size_t elem) {
    const size_t image = kRefBytes + elem * 2 * kMfRows * kPitch;
    const size_t finish = sizeof(double) * (4 * kFrames + kFrames + 16 * kM);
    return image > finish ? image : finish;
  }
};

template <typename T, int kM>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 8 && kM == 256) ? 1 : 2)))
void msd_band_mfma_kernel(
    const T *__restrict__ x, int64_t n_series, int64_t n_frames, int64_t t0, int64_t t_len,
    int R, int64_t series_per_chunk, double *__restrict__ partial, int64_t tiles_t,
    int tiles_m, int n_chunks) {
  using C = MsdMf<kM>;
  int64_t block_t;
  int block_m, block_chunk;
  if (!band_item(tiles_t, tiles_m, n_chunks, &block_t, &block_m, &block_chunk)) return;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double(*refs)[kMfRows] = reinterpret_cast<double(*)[kMfRows]>(lds_raw);
  T(*img)[kMfRows][C::kPitch] =
      reinterpret_cast<T(*)[kMfRows][C::kPitch]>(lds_raw + C::kRefBytes);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t tile_t = block_t * C::kT;
  const int tile_m = block_m * kM;
  const int64_t s0 = static_cast<int64_t>(block_chunk) * series_per_chunk;
  const int64_t s1 = min(n_series, s0 + series_per_chunk);

  constexpr int kSlots = kMfSeries / 4;
  int64_t own_frame[C::kPerLane];
#pragma unroll
  for (int q = 0; q < C::kPerLane; ++q) {
    const int f = lane + 64 * q;
    const int64_t frame =
        (f < C::kT) ? (t0 + tile_t + f) : (t0 + tile_t + tile_m + (f - C::kT));
    own_frame[q] = (f < C::kFrames && frame < n_frames) ? frame : -1;
  }
  T pre[kSlots][C::kPerLane][3];
  auto fetch = [&](int64_t s_base) {
#pragma unroll
    for (int u = 0; u < kSlots; ++u) {
      const int64_t s = s_base + wave + 4 * u;
      const T *series = x + s * n_frames * 3;
#pragma unroll
      for (int q = 0; q < C::kPerLane; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          pre[u][q][d] = (s < s1 && own_frame[q] >= 0) ? series[own_frame[q] * 3 + d] : T(0);
    }
  };

  v4d acc[C::kTiles];
#pragma unroll
  for (int j = 0; j < C::kTiles; ++j) acc[j] = v4d{0.0, 0.0, 0.0, 0.0};
  double sq[C::kPerLane];
#pragma unroll
  for (int q = 0; q < C::kPerLane; ++q) sq[q] = 0.0;

  if (s0 < s1) fetch(s0);
  int buf = 0;
  const int kk = lane >> 4;
  const int col = lane & 15;
  for (int64_t s_base = s0; s_base < s1; s_base += kMfSeries) {
#pragma unroll
    for (int u = 0; u < kSlots; ++u)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        // lane 0 holds the tile's first frame of this wave's series
        const double ref = static_cast<double>(__shfl(pre[u][0][d], 0));
        if (lane == 0) refs[buf][(wave + 4 * u) * 3 + d] = ref;
#pragma unroll
        for (int q = 0; q < C::kPerLane; ++q) {
          img[buf][(wave + 4 * u) * 3 + d][lane + 64 * q] = pre[u][q][d];
          const double diff = static_cast<double>(pre[u][q][d]) - ref;
          sq[q] = fma(diff, diff, sq[q]);
        }
      }
    __syncthreads();
    if (s_base + kMfSeries < s1) fetch(s_base + kMfSeries);
#pragma unroll
    for (int g = 0; g < kMfRows / 4; ++g) {
      const T *row = img[buf][4 * g + kk];
      const double ref = refs[buf][4 * g + kk];
      const double a = static_cast<double>(row[16 * wave + col]) - ref;
#pragma unroll
      for (int j = 0; j < C::kTiles; ++j) {
        const double b = static_cast<double>(row[C::kT + 16 * (wave + j) + col]) - ref;
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      }
    }
    buf ^= 1;  // the other image was last read before this round's barrier
  }

  // per staged frame: squares summed over the four waves' series
  __syncthreads();  // the images are no longer read
  double *sqs = reinterpret_cast<double *>(lds_raw);  // [4][kFrames]
  double *ssum = sqs + 4 * C::kFrames;                // [kFrames]
  double *strip = ssum + C::kFrames;                  // [4 waves x 4 rows][kM]
#pragma unroll
  for (int q = 0; q < C::kPerLane; ++q)
    if (lane + 64 * q < C::kFrames) sqs[wave * C::kFrames + lane + 64 * q] = sq[q];
  __syncthreads();
  for (int f = threadIdx.x; f < C::kFrames; f += kThreads)
    ssum[f] = ((sqs[f] + sqs[C::kFrames + f]) + sqs[2 * C::kFrames + f]) + sqs[3 * C::kFrames + f];
  // cross terms: lag m collects the diagonal D[t][t + m] of every wave's strip, four
  // rows per wave at a time, in a fixed order
  double cross = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    __syncthreads();  // ssum written / the previous rows consumed
    const int trow = kk + 4 * r;
#pragma unroll
    for (int j = 0; j < C::kTiles; ++j) {
      const int m_rel = 16 * j + col - trow;
      if (m_rel >= 0 && m_rel < kM) strip[(wave * 4 + kk) * kM + m_rel] = acc[j][r];
    }
    __syncthreads();
    if (threadIdx.x < kM) {
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
          if (tile_t + 16 * w + k4 + 4 * r < t_len)
            cross = cross + strip[(w * 4 + k4) * kM + threadIdx.x];
    }
  }
  const int m = tile_m + static_cast<int>(threadIdx.x);
  if (threadIdx.x < kM && m < R) {
    double own = 0.0, shifted = 0.0;
    for (int t = 0; t < C::kT; ++t)
      if (tile_t + t < t_len) {
        own = own + ssum[t];
        shifted = shifted + ssum[C::kT + t + threadIdx.x];
      }
    const double value = (own + shifted) - 2.0 * cross;
    partial[(static_cast<int64_t>(block_chunk) * tiles_t + block_t) * R + m] =
        (m == 0) ? 0.0 : value;
  }
}

// The same band on v_mfma_f64_4x4x4_4b_f64, which sustains 74 TFLOP/s on gfx950 where
// the 16x16x4 instruction stops at 48 (scratch/mfma/clock.hip).  One instruction
// multiplies four independent 4 x 4 x 4 blocks; lane l supplies A[blk][i][k] and
// B[blk][k][j] with k = l / 16, blk = (l % 16) / 4, i or j = l % 4, and receives
// D[blk][i][j] with i = l / 16, blk = (l % 16) / 4, j = l % 4 (scratch/mfma/probe4.hip).
// An operand register therefore holds 16 consecutive frames of 4 rows, as for the larger
// instruction: with A = own frames f.. and B = frames g.., block blk forms the products of
// own frames f + 4 blk + i with frames g + 4 blk + j, i.e. the lags g - f + (j - i).
// A wave owns 16 own frames (one A register per inner block) and walks B over the
// shifted frames in steps of 4: b_u, frames 16 wave + 4 u.., gives the lags 4 u - 3 ..
// 4 u + 3, accumulator acc[u] (one double per lane).  Every (own frame, lag) pair lands
// in exactly one accumulator, so a tile of 128 frames x 4 (NQ - 1) lags costs 128 x 4 NQ
// products per row: NQ = 27 covers 104 lags at 96 %.  Blocks have eight waves and a
// compute unit holds two of them: four waves per SIMD keep the matrix pipe fed while
// others stage or wait at the barrier (with two it idled 29 % of the time).
// The image is staged as float64 (four series per round), so an operand goes from LDS
// to the instruction with no conversion in between.  kMsd as in msd_band_mfma_kernel
// (reference position per series and tile, here subtracted when the image is written;
// squares by the staging threads; lag 0 exact); otherwise the products go to the band
// table Q.
constexpr int kB4Threads = 512;
constexpr int kB4Series = 4;  // series per staging round
constexpr int kB4Rows = 3 * kB4Series;
template <int NQ>
struct Band4 {
  static constexpr int kT = (kB4Threads / 64) * 16;  // own frames per block
  static constexpr int kM = 4 * (NQ - 1);            // lags per tile
  static constexpr int kFrames = kT + (kT + kM);
  static constexpr int kPerLane = (kFrames + 127) / 128;  // frames a thread stages per series
  // ds_read_b64 serves lanes 0-31 in one cycle over 64 four-byte banks: the two rows of a
  // half wave must start 32 banks (16 doubles) apart
  static constexpr int kPitch = 128 * kPerLane + 16;
  static constexpr int kStripPitch = kM + 1;
  // the image is float64 whatever the input: operands go from LDS to the instruction
  static size_t lds_bytes(bool msd) {
    const size_t image = sizeof(double) * 2 * kB4Rows * kPitch;
    const size_t finish = msd ? sizeof(double) * (4 * kFrames + kFrames + 32 * kStripPitch) : 0;
    return image > finish ? image : finish;
  }
};

template <typename T, int NQ, bool kMsd>
__global__ __launch_bounds__(kB4Threads) __attribute__((amdgpu_waves_per_eu(4)))
void band4_kernel(const T *__restrict__ x, int64_t n_series, int64_t n_frames, int64_t t0,
                  int64_t t_len, int R, int64_t series_per_chunk, double *__restrict__ Q,
                  int64_t tiles_t, int tiles_m, int n_chunks) {
  using C = Band4<NQ>;
  int64_t block_t;
  int block_m, block_chunk;
  if (!band_item(tiles_t, tiles_m, n_chunks, &block_t, &block_m, &block_chunk)) return;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double(*img)[kB4Rows][C::kPitch] = reinterpret_cast<double(*)[kB4Rows][C::kPitch]>(lds_raw);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t tile_t = block_t * C::kT;
  const int tile_m = block_m * C::kM;
  const int64_t s0 = static_cast<int64_t>(block_chunk) * series_per_chunk;
  const int64_t s1 = min(n_series, s0 + series_per_chunk);

  // Staging of float32 input: a pair of waves takes one series of the round, a lane four
  // adjacent staged frames (the tile's own frames first, then the kT + kM frames from the
  // tile's first lag on): 48 contiguous bytes, three 16-byte buffer loads, and per
  // component four adjacent doubles of the image, two 16-byte stores.  Memory
  // instructions next to the matrix instructions cost the loop far more than their bytes
  // (scratch/mfma/lds_mfma.hip), so a round issues 18 + 36 of them per block instead of
  // 48 + 48.  The buffer descriptor covers one series: frames past its end, the lanes
  // past the last group and series past the chunk read as zero.
  constexpr bool kQuad = sizeof(T) == 4;
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u3v __attribute__((ext_vector_type(3)));
  typedef double d2v __attribute__((ext_vector_type(2)));
  const int q_series = __builtin_amdgcn_readfirstlane(wave >> 1);
  const int q_group = (wave & 1) * 64 + lane;
  const bool q_active = q_group < C::kFrames / 4;
  const int q_f = 4 * q_group;
  const unsigned series_bytes = static_cast<unsigned>(n_frames * 12);
  const unsigned q_offset =
      q_active ? static_cast<unsigned>(
                     (t0 + tile_t + ((q_f < C::kT) ? q_f : tile_m + (q_f - C::kT))) * 12)
               : series_bytes;
  u4v pre4[3];
  u3v ref3;
  // Staging of float64 input: wave % 4 = series of the round, wave / 4 = half of the staged
  // frames, lane = frames lane + 64 q.
  const int st_series = wave & 3;
  const int st_first = (wave >> 2) * 64 * C::kPerLane + lane;
  T pre[C::kPerLane][3];
  T pre_ref[3];
  auto fetch = [&](int64_t s_base) {
    if constexpr (kQuad) {
      const int64_t s = s_base + q_series;
      const bool there = s < s1;
      const T *series = x + (there ? s : s0) * n_frames * 3;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<T *>(series), 0, there ? static_cast<int>(series_bytes) : 0, 0x00020000);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        pre4[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(q_offset) + 16 * j,
                                                        0, 0);
      if (kMsd)  // the series' position at the tile's first frame (one address per wave)
        ref3 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, static_cast<int>((t0 + tile_t) * 12),
                                                    0, 0);
    } else {
      const int64_t s = s_base + st_series;
      const T *series = x + s * n_frames * 3;
#pragma unroll
      for (int q = 0; q < C::kPerLane; ++q) {
        const int f = st_first + 64 * q;
        const int64_t frame = t0 + tile_t + ((f < C::kT) ? f : tile_m + (f - C::kT));
        const bool live = s < s1 && f < C::kFrames && frame < n_frames;
#pragma unroll
        for (int d = 0; d < 3; ++d) pre[q][d] = live ? series[frame * 3 + d] : T(0);
      }
      if (kMsd) {
        // the series' position at the tile's first frame (one address per wave)
#pragma unroll
        for (int d = 0; d < 3; ++d) pre_ref[d] = (s < s1) ? series[(t0 + tile_t) * 3 + d] : T(0);
      }
    }
  };

  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  static_assert(C::kPerLane <= 4, "sq holds four staged frames per lane");
  double sq[4] = {0.0, 0.0, 0.0, 0.0};

  if (s0 < s1) fetch(s0);
  int buf = 0;
  const int kk = lane >> 4;
  const int fr = lane & 15;
  for (int64_t s_base = s0; s_base < s1; s_base += kB4Series) {
    if constexpr (kQuad) {
      if (q_active) {
        const unsigned word[12] = {pre4[0].x, pre4[0].y, pre4[0].z, pre4[0].w,
                                   pre4[1].x, pre4[1].y, pre4[1].z, pre4[1].w,
                                   pre4[2].x, pre4[2].y, pre4[2].z, pre4[2].w};
        const unsigned ref_word[3] = {ref3.x, ref3.y, ref3.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const double ref = kMsd ? static_cast<double>(__uint_as_float(ref_word[d])) : 0.0;
          double v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = static_cast<double>(__uint_as_float(word[3 * i + d])) - ref;
            if (kMsd) sq[i] = fma(v[i], v[i], sq[i]);
          }
          d2v *dst = reinterpret_cast<d2v *>(&img[buf][q_series * 3 + d][q_f]);
          dst[0] = d2v{v[0], v[1]};
          dst[1] = d2v{v[2], v[3]};
        }
      }
    } else {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double ref = kMsd ? static_cast<double>(pre_ref[d]) : 0.0;
#pragma unroll
        for (int q = 0; q < C::kPerLane; ++q) {
          const double v = static_cast<double>(pre[q][d]) - ref;
          img[buf][st_series * 3 + d][st_first + 64 * q] = v;
          if (kMsd) sq[q] = fma(v, v, sq[q]);
        }
      }
    }
    __syncthreads();
    if (s_base + kB4Series < s1) fetch(s_base + kB4Series);
    // The round's products as one stream of 3 NQ steps (inner block g = step / NQ, lag
    // block u = step % NQ).  B operands are read kAhead steps before their use, across the
    // inner blocks' edges (all waves reach those together: operands requested only there
    // would leave the matrix pipe idle for an LDS latency three more times per round); the
    // scheduling fence keeps the compiler from hoisting more reads (and spilling
    // accumulators to make room).
    constexpr int kAhead = 6;
    constexpr int kSteps = (kB4Rows / 4) * NQ;
    const double *own = &img[buf][kk][16 * wave + fr];
    const double *shifted = own + C::kT;
    auto operand_b = [&](int step) {
      return shifted[(step / NQ) * 4 * C::kPitch + 4 * (step % NQ)];
    };
    double a[kB4Rows / 4];
    double b[kSteps];
    a[0] = own[0];
#pragma unroll
    for (int step = 0; step < kAhead; ++step) b[step] = operand_b(step);
#pragma unroll
    for (int step = 0; step < kSteps; ++step) {
      if (step + kAhead < kSteps) {
        b[step + kAhead] = operand_b(step + kAhead);
        if ((step + kAhead) % NQ == 0)
          a[(step + kAhead) / NQ] = own[((step + kAhead) / NQ) * 4 * C::kPitch];
      }
      // keeps the previous step's operand alive past this read, so that the read lands
      // in a register whose last instruction was issued two steps ago, not just now
      if (step > 0) asm volatile("" ::"v"(b[step - 1]));
      acc[step % NQ] =
          __builtin_amdgcn_mfma_f64_4x4x4f64(a[step / NQ], b[step], acc[step % NQ], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    buf ^= 1;  // the other image was last read before this round's barrier
  }

  // result lane layout: own frame 4 blk + i, lag 4 u + j - i
  const int ri = lane >> 4;
  const int rblk = (lane & 15) >> 2;
  const int rj = lane & 3;
  if (!kMsd) {
    double *out = Q + static_cast<int64_t>(block_chunk) * t_len * R;
    const int64_t tt = tile_t + 16 * wave + 4 * rblk + ri;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m_rel = 4 * q + rj - ri;
      const int m = tile_m + m_rel;
      if (m_rel >= 0 && m_rel < C::kM && m < R && tt < t_len) out[tt * R + m] = acc[q];
    }
    return;
  }

  // per staged frame: squares summed over the round's four series
  __syncthreads();  // the images are no longer read
  double *sqs = reinterpret_cast<double *>(lds_raw);  // [4][kFrames]
  double *ssum = sqs + 4 * C::kFrames;                // [kFrames]
  double *strip = ssum + C::kFrames;                  // [32 own frames][kStripPitch]
  if constexpr (kQuad) {
    if (q_active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sqs[q_series * C::kFrames + q_f + i] = sq[i];
    }
  } else {
#pragma unroll
    for (int q = 0; q < C::kPerLane; ++q)
      if (st_first + 64 * q < C::kFrames) sqs[st_series * C::kFrames + st_first + 64 * q] = sq[q];
  }
  __syncthreads();
  for (int f = threadIdx.x; f < C::kFrames; f += kB4Threads)
    ssum[f] = ((sqs[f] + sqs[C::kFrames + f]) + sqs[2 * C::kFrames + f]) + sqs[3 * C::kFrames + f];
  // cross terms: lag m collects its diagonal, two waves' 32 own frames at a time, in a
  // fixed order
  double cross = 0.0;
#pragma unroll
  for (int pass = 0; pass < kB4Threads / 128; ++pass) {
    __syncthreads();  // ssum written / the previous rows consumed
    if ((wave >> 1) == pass) {
      double *mine = strip + ((wave & 1) * 16 + 4 * rblk + ri) * C::kStripPitch;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int m_rel = 4 * q + rj - ri;
        if (m_rel >= 0 && m_rel < C::kM) mine[m_rel] = acc[q];
      }
    }
    __syncthreads();
    if (threadIdx.x < C::kM) {
      for (int r = 0; r < 32; ++r)
        if (tile_t + 32 * pass + r < t_len) cross = cross + strip[r * C::kStripPitch + threadIdx.x];
    }
  }
  const int m = tile_m + static_cast<int>(threadIdx.x);
  if (threadIdx.x < C::kM && m < R) {
    double own_sum = 0.0, shifted_sum = 0.0;
    for (int t = 0; t < C::kT; ++t)
      if (tile_t + t < t_len) {
        own_sum = own_sum + ssum[t];
        shifted_sum = shifted_sum + ssum[C::kT + t + threadIdx.x];
      }
    const double value = (own_sum + shifted_sum) - 2.0 * cross;
    Q[(static_cast<int64_t>(block_chunk) * tiles_t + block_t) * R + m] = (m == 0) ? 0.0 : value;
  }
}

// ------------------------------------------------------------------------ MSD

// partial[block][i] = sum over the block's windows and series of
//   sum_d (x[s][w*ct + tau[i]][d] - x[s][w*ct][d])^2
template <typename T>
__global__ __launch_bounds__(kThreads) void msd_kernel(
    const T *__restrict__ x, int64_t n_series, int64_t n_frames, int k, int64_t ct,
    int64_t win_begin, int64_t n_win, int wpb, const int32_t *__restrict__ tau, int R, int D,
    int64_t series_per_chunk, double *__restrict__ partial) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *img = reinterpret_cast<double *>(lds_raw);
  const int64_t wl0 = static_cast<int64_t>(blockIdx.x) * wpb;
  const int nw = static_cast<int>((n_win - wl0) < wpb ? (n_win - wl0) : wpb);
  const int64_t frame0 = (win_begin + wl0) * ct;
  const int span = static_cast<int>((nw - 1) * ct) + D;
  const int64_t s0 = static_cast<int64_t>(blockIdx.y) * series_per_chunk;
  const int64_t s1 = min(n_series, s0 + series_per_chunk);
  double *out = partial + (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * R;
  // each thread owns the indices i = tid, tid + 256, ... of `out`
  for (int i = threadIdx.x; i < R; i += kThreads) out[i] = 0.0;
  for (int64_t s = s0; s < s1; ++s) {
    __syncthreads();
    stage_span(x + s * n_frames * k, frame0, span, k, img);
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += kThreads) {
      const int t = tau[i];
      double sum = 0.0;
      for (int wl = 0; wl < nw; ++wl) {
        const double *origin = img + static_cast<size_t>(wl) * ct * k;
        const double *later = origin + static_cast<size_t>(t) * k;
        for (int d = 0; d < k; ++d) {
          const double diff = later[d] - origin[d];
          sum = fma(diff, diff, sum);
        }
      }
      out[i] = out[i] + sum;
    }
  }
}

// out[i] = scale * sum_b partial[b][i].  One wave per lag: lane l adds blocks l, l + 64,
// ... in order, then the 64 partial sums are combined by a fixed butterfly.
__global__ __launch_bounds__(kThreads) void combine_blocks_kernel(
    const double *__restrict__ partial, int64_t n_blocks, int R, double scale,
    double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (i >= R) return;
  double s = 0.0;
  for (int64_t b = lane; b < n_blocks; b += 64) s = s + partial[b * R + i];
  for (int off = 32; off > 0; off >>= 1) s = s + __shfl_xor(s, off);
  if (lane == 0) out[i] = s * scale;
}

// ----------------------------------------------------------------- host side

// Series chunks of a band launch with `tiles` blocks per chunk: at least `base` of them
// (what fills the chip), and then the smallest count up to `most` whose last round of
// resident blocks is at least 95 % full, or the fullest one (3 128 tiles on 512 resident
// blocks run 7 rounds for 6.1 rounds of work; three chunks run 19 for 18.3).
int64_t band_chunks(int64_t tiles, int64_t base, int64_t most) {
  int cus = 256;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  const int64_t resident = 2 * static_cast<int64_t>(cus);
  if (base < 1) base = 1;
  if (most < base) most = base;
  int64_t best = base;
  double best_fill = 0.0;
  for (int64_t c = base; c <= most; ++c) {
    const int64_t blocks = tiles * c;
    const double fill = static_cast<double>(blocks) /
                        static_cast<double>(ceil_div(blocks, resident) * resident);
    if (fill > best_fill + 1e-9) {
      best_fill = fill;
      best = c;
    }
    if (fill >= 0.95) return c;
  }
  return best;
}

struct TauInfo {
  int D = 0;
  bool identity = true;
};

int check_windows(const void *x, int dtype, int location, int64_t n_series,
                  int64_t n_frames, int k, int64_t ct, int64_t win_begin, int64_t win_end,
                  const int32_t *tau, int n_tau, TauInfo *info) {
  MDX_REQUIRE(x != nullptr && tau != nullptr, "null argument");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  MDX_REQUIRE(n_series >= 1 && n_frames >= 1, "empty input");
  MDX_REQUIRE(k >= 1 && k <= 16, "k out of range");
  MDX_REQUIRE(ct >= 1, "correlation_time must be >= 1");
  MDX_REQUIRE(n_tau >= 1 && n_tau <= 8192, "n_tau out of range (1..8192)");
  MDX_REQUIRE(win_begin >= 0 && win_end >= win_begin, "bad window range");
  int mx = 0;
  bool ident = true;
  for (int i = 0; i < n_tau; ++i) {
    MDX_REQUIRE(tau[i] >= 0, "negative tau value");
    if (tau[i] > mx) mx = tau[i];
    if (tau[i] != i) ident = false;
  }
  info->D = mx + 1;
  info->identity = ident;
  if (win_end > win_begin) {
    const int64_t last = (win_end - 1) * ct + info->D;
    MDX_REQUIRE(last <= n_frames,
                "window %lld needs frames up to %lld but the batch holds %lld",
                (long long)(win_end - 1), (long long)last, (long long)n_frames);
  }
  return MDX_OK;
}

// windows per block such that the staged image stays within `budget` bytes
int windows_per_block(int64_t ct, int D, int k, int64_t n_win, int cap, size_t budget) {
  int64_t wpb = cap;
  if (wpb > n_win) wpb = n_win;
  while (wpb > 1 && static_cast<size_t>((wpb - 1) * ct + D) * k * sizeof(double) > budget) --wpb;
  if (wpb < 1) wpb = 1;
  return static_cast<int>(wpb);
}

// Inputs of the sliding-window path.  `series` [n_series][n_frames][k] feeds the lag
// products directly (band == false) or through the band table (band == true; may be
// null with cross_a set and no self term).  With cross_a / cross_b set the products
// are cross_weight * a[t+m].b[t] - self_weight * (band table of `series`).
struct SlideJob {
  int dtype = MDX_F64;
  const void *series = nullptr;
  int64_t n_series = 0, n_frames = 0;
  int k = 3;
  int64_t ct = 1, win_begin = 0, n_win = 0;
  int R = 0;
  double scale = 1.0, integral_scale = 1.0;
  const double *d_time = nullptr;
  bool band = false;
  bool lag_normalise = true;
  const void *cross_a = nullptr, *cross_b = nullptr;
  double cross_weight = 1.0, self_weight = 0.0;
};

// Runs acf_slide_kernel over the job's windows (in segments when a band table is
// needed); per-block results land in d_acf / d_mean / d_m2, block_windows gets the
// number of windows of every block.
int slide_windows(const SlideJob &job, hipStream_t stream, DeviceBuffer &d_C,
                  DeviceBuffer &d_acf, DeviceBuffer &d_mean, DeviceBuffer &d_m2,
                  std::vector<int64_t> &block_windows) {
  const int dtype = job.dtype;
  const void *dx = job.series;
  const int64_t n_series = job.n_series, n_frames = job.n_frames;
  const int k = job.k;
  const int64_t correlation_time = job.ct, win_begin = job.win_begin, n_win = job.n_win;
  const int R = job.R;
  const double scale = job.scale, integral_scale = job.integral_scale;
  const int lag_normalise = job.lag_normalise ? 1 : 0;
  const int n_int = R > 1 ? R - 1 : 1;
  const bool band = job.band;
  const bool cross = job.cross_a != nullptr;
  {
    // four lags per thread at most before the block grows to 1024 threads
    // one series straight from the caller's array: the strip kernel
    const bool strip = !band && !cross && n_series == 1;
    const int strip_waves = R > 256 ? 8 : (R > 128 ? 4 : 2);
    const int slide_block = (R > 1024) ? 1024 : 256;
    const size_t lds_s = sizeof(double) * (2 * static_cast<size_t>(R) + slide_block / 64);
    static_assert(8192 <= kSlidePerMax * 1024, "lags per thread");
    // lags per thread, rounded up to an instantiated count
    const int per_need = static_cast<int>(ceil_div(R, slide_block));
    const int slide_per = slide_block == 256 ? 4 : (per_need <= 3 ? 3 : per_need <= 5 ? 5 : 8);
    // windows per block: enough to amortise the directly computed first window
    // (R/2 products per lag against 2*ct per slide), few enough to fill the chip
    // and to bound the drift of the running sums
    auto plan_wpb = [&](int64_t windows) {
      int64_t amortise = R / (4 * correlation_time);
      const int64_t spread = ceil_div(windows, 256);
      if (amortise > spread) amortise = spread;
      int64_t wpb = ceil_div(windows, 2048);
      if (wpb < amortise) wpb = amortise;
      // the cap bounds the drift of the running sums; long windows need more
      // slides to pay for their first window
      const int64_t cap = R > 512 ? R : 512;
      if (wpb > cap) wpb = cap;
      if (wpb < 1) wpb = 1;
      if (strip) {
        // whole rounds of resident blocks (two per compute unit), whole strips
        const int64_t slots = 512;
        int64_t blocks = ceil_div(windows, wpb);
        if (blocks > slots) {
          blocks = blocks / slots * slots;
          wpb = ceil_div(windows, blocks);
        }
        wpb = ceil_div(wpb, kStripWin) * kStripWin;
      }
      return wpb;
    };
    // window segments: the whole range at once from the series, or as many windows
    // as a 2 GiB band table covers
    int64_t seg_windows = n_win;
    int64_t n_chunks = 1, series_per_chunk = n_series;
    if (band) {
      // series chunks only as far as needed to fill the chip with band tiles
      const bool b4 = k == 3 && getenv("MDX_ACF_BAND_VALU") == nullptr &&
                      getenv("MDX_BAND_MFMA16") == nullptr &&
                      (dtype != MDX_F32 || n_frames * 12 < (int64_t(1) << 31));
      const int64_t tiles =
          b4 ? ceil_div((n_win - 1) * correlation_time + R, Band4<33>::kT) *
                   ceil_div(R, Band4<33>::kM)
             : ceil_div((n_win - 1) * correlation_time + R, kMfT) * ceil_div(R, kMfM);
      int64_t most = n_series / 32 < 16 ? n_series / 32 : 16;
      // every chunk has its own plane of the table: stay within half the budget
      const int64_t plane_bytes = ((n_win - 1) * correlation_time + R) * R * 8;
      while (most > 1 && plane_bytes * most > (int64_t(1) << 30)) --most;
      n_chunks = band_chunks(tiles, ceil_div(2048, tiles) < most ? ceil_div(2048, tiles) : most,
                             most);
      if (n_chunks < 1) n_chunks = 1;
      series_per_chunk = ceil_div(n_series, n_chunks);
      n_chunks = ceil_div(n_series, series_per_chunk);
      // table budget: 2 GiB unless MDX_ACF_BAND_TABLE_BYTES says otherwise
      size_t budget = size_t(2) << 30;
      if (const char *env = getenv("MDX_ACF_BAND_TABLE_BYTES")) {
        const long long v = atoll(env);
        if (v > 0) budget = static_cast<size_t>(v);
      }
      const int64_t max_frames =
          static_cast<int64_t>(budget / (sizeof(double) * R * n_chunks));
      MDX_REQUIRE(max_frames > R, "data_range %d is too large for the band table", R);
      seg_windows = (max_frames - R) / correlation_time + 1;
      if (seg_windows > n_win) seg_windows = n_win;
    }
    int64_t total_blocks = 0;
    for (int64_t ws = 0; ws < n_win; ws += seg_windows) {
      const int64_t nw = std::min(seg_windows, n_win - ws);
      total_blocks += ceil_div(nw, plan_wpb(nw));
    }
    MDX_TRY(d_acf.alloc(sizeof(double) * total_blocks * R));
    MDX_TRY(d_mean.alloc(sizeof(double) * total_blocks * n_int));
    MDX_TRY(d_m2.alloc(sizeof(double) * total_blocks * n_int));
    const int64_t seg_frames = (seg_windows - 1) * correlation_time + R;
    if (band) MDX_TRY(d_C.alloc(sizeof(double) * n_chunks * seg_frames * R));
    const double *d_time_ptr = job.d_time;
    // per-lag weights; the buffer goes back to the pool when this function returns and
    // can only be handed to later work of the same stream
    DeviceBuffer d_weight, d_planes, d_coef, d_state;
    MDX_TRY(d_weight.alloc(sizeof(double) * R));
    hipLaunchKernelGGL(lag_weight_kernel, dim3(static_cast<unsigned>(ceil_div(R, kThreads))),
                       dim3(kThreads), 0, stream, d_weight.as<double>(), R, scale,
                       lag_normalise);
    MDX_HIP(hipGetLastError());

    int64_t block_offset = 0;
    for (int64_t ws = 0; ws < n_win; ws += seg_windows) {
      const int64_t nw = std::min(seg_windows, n_win - ws);
      const int64_t wpb = plan_wpb(nw);
      const int64_t n_blocks = ceil_div(nw, wpb);
      for (int64_t b = 0; b < n_blocks; ++b)
        block_windows.push_back(std::min(wpb, nw - b * wpb));
      const int64_t t0 = (win_begin + ws) * correlation_time;
      const int64_t t_len = (nw - 1) * correlation_time + R;
      if (band) {
        using Tile = BandTile<32>;
        const size_t lds_q = Tile::lds_bytes(k);
        const int64_t q_tiles_t = ceil_div(t_len, Tile::T), q_tiles_m = ceil_div(R, Tile::M);
        const int64_t mf_tiles_t = ceil_div(t_len, kMfT), mf_tiles_m = ceil_div(R, kMfM);
        dim3 grid_q(band_grid(q_tiles_t, q_tiles_m, n_chunks));
        dim3 grid_mf(band_grid(mf_tiles_t, mf_tiles_m, n_chunks));
        KernelTimer timer(MDX_SLOT_ACF_PARTIAL, stream);
#define MDX_LAUNCH_BAND(TYPE, KK)                                                            \
  do {                                                                                       \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(acf_band_kernel<TYPE, KK, 32, false>),                \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_q)));               \
    hipLaunchKernelGGL((acf_band_kernel<TYPE, KK, 32, false>), grid_q, dim3(kThreads), lds_q, \
                       stream, static_cast<const TYPE *>(dx), n_series, n_frames, k, t0,     \
                       t_len, R, series_per_chunk, d_C.as<double>(), q_tiles_t,              \
                       static_cast<int>(q_tiles_m), static_cast<int>(n_chunks));             \
  } while (0)
#define MDX_LAUNCH_BAND_MFMA(TYPE)                                                           \
  do {                                                                                       \
    const size_t lds_mf = sizeof(TYPE) * 2 * kMfRows * kMfPitch;                             \
    MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(acf_band_mfma_kernel<TYPE>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize,                  \
                                static_cast<int>(lds_mf)));                                  \
    hipLaunchKernelGGL((acf_band_mfma_kernel<TYPE>), grid_mf, dim3(kThreads), lds_mf,        \
                       stream, static_cast<const TYPE *>(dx), n_series, n_frames, t0, t_len, \
                       R, series_per_chunk, d_C.as<double>(), mf_tiles_t,                    \
                       static_cast<int>(mf_tiles_m), static_cast<int>(n_chunks));            \
  } while (0)
#define MDX_LAUNCH_BAND4(TYPE)                                                                \
  do {                                                                                       \
    using B4 = Band4<33>;                                                                    \
    const size_t lds_b4 = B4::lds_bytes(false);                                              \
    const int64_t b4_tiles_t = ceil_div(t_len, B4::kT), b4_tiles_m = ceil_div(R, B4::kM);    \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(band4_kernel<TYPE, 33, false>),                       \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b4)));              \
    hipLaunchKernelGGL((band4_kernel<TYPE, 33, false>),                                      \
                       dim3(band_grid(b4_tiles_t, b4_tiles_m, n_chunks)), dim3(kB4Threads),  \
                       lds_b4, stream, static_cast<const TYPE *>(dx), n_series, n_frames,    \
                       t0, t_len, R, series_per_chunk, d_C.as<double>(), b4_tiles_t,         \
                       static_cast<int>(b4_tiles_m), static_cast<int>(n_chunks));            \
  } while (0)
        const bool mfma = (k == 3) && getenv("MDX_ACF_BAND_VALU") == nullptr;
        const bool mfma4 = mfma && getenv("MDX_BAND_MFMA16") == nullptr &&
                           (dtype != MDX_F32 || n_frames * 12 < (int64_t(1) << 31));
        if (mfma4) {
          if (dtype == MDX_F32) MDX_LAUNCH_BAND4(float); else MDX_LAUNCH_BAND4(double);
        } else if (dtype == MDX_F32) {
          if (mfma) MDX_LAUNCH_BAND_MFMA(float);
          else if (k == 3) MDX_LAUNCH_BAND(float, 3);
          else MDX_LAUNCH_BAND(float, 0);
        } else {
          if (mfma) MDX_LAUNCH_BAND_MFMA(double);
          else if (k == 3) MDX_LAUNCH_BAND(double, 3);
          else MDX_LAUNCH_BAND(double, 0);
        }
#undef MDX_LAUNCH_BAND4
#undef MDX_LAUNCH_BAND_MFMA
#undef MDX_LAUNCH_BAND
        MDX_HIP(hipGetLastError());
        if (n_chunks > 1) {
          const int64_t plane = t_len * R;
          hipLaunchKernelGGL(band_combine_kernel,
                             dim3(static_cast<unsigned>(ceil_div(plane, kThreads))),
                             dim3(kThreads), 0, stream, d_C.as<double>(),
                             static_cast<int>(n_chunks), plane);
          MDX_HIP(hipGetLastError());
        }
      }
      KernelTimer timer(MDX_SLOT_ACF_FINISH, stream);
#define MDX_LAUNCH_SLIDE_B(PRODUCT_TYPE, PRODUCT, BLOCK, PER)                                 \
  do {                                                                                       \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(acf_slide_kernel<PRODUCT_TYPE, BLOCK, PER>),          \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_s)));               \
    hipLaunchKernelGGL((acf_slide_kernel<PRODUCT_TYPE, BLOCK, PER>),                         \
                       dim3(static_cast<unsigned>(n_blocks)), dim3(BLOCK), lds_s, stream,    \
                       PRODUCT, correlation_time, win_begin + ws, nw, wpb, R,                \
                       d_weight.as<double>(), integral_scale, d_time_ptr, block_offset,      \
                       d_acf.as<double>(), d_mean.as<double>(), d_m2.as<double>());          \
  } while (0)
#define MDX_LAUNCH_SLIDE(PRODUCT_TYPE, PRODUCT)                                              \
  do {                                                                                       \
    if (slide_block == 256)                                                                  \
      MDX_LAUNCH_SLIDE_B(PRODUCT_TYPE, PRODUCT, 256, 4);                                     \
    else if (slide_per == 3)                                                                 \
      MDX_LAUNCH_SLIDE_B(PRODUCT_TYPE, PRODUCT, 1024, 3);                                    \
    else if (slide_per == 5)                                                                 \
      MDX_LAUNCH_SLIDE_B(PRODUCT_TYPE, PRODUCT, 1024, 5);                                    \
    else                                                                                     \
      MDX_LAUNCH_SLIDE_B(PRODUCT_TYPE, PRODUCT, 1024, 8);                                    \
  } while (0)
#define MDX_LAUNCH_SLIDE_SERIES(TYPE)                                                        \
  do {                                                                                       \
    const TYPE *xs = static_cast<const TYPE *>(dx);                                          \
    if (k == 3 && n_series == 1) {                                                           \
      using P = SeriesProduct<TYPE, 3, true>;                                                \
      P prod{xs, n_series, n_frames, k};                                                     \
      MDX_LAUNCH_SLIDE(P, prod);                                                             \
    } else if (k == 3) {                                                                     \
      using P = SeriesProduct<TYPE, 3, false>;                                               \
      P prod{xs, n_series, n_frames, k};                                                     \
      MDX_LAUNCH_SLIDE(P, prod);                                                             \
    } else {                                                                                 \
      using P = SeriesProduct<TYPE, 0, false>;                                               \
      P prod{xs, n_series, n_frames, k};                                                     \
      MDX_LAUNCH_SLIDE(P, prod);                                                             \
    }                                                                                        \
  } while (0)
      if (strip) {
        const int lags_per_phase = strip_waves * 64;
        const int padded = static_cast<int>(ceil_div(R, lags_per_phase)) * lags_per_phase;
        const int64_t plane_pitch = t_len + kPlaneFront + kPlaneBack;
        const size_t plane_bytes = dtype_size(dtype) * static_cast<size_t>(plane_pitch) * k;
        MDX_TRY(d_planes.alloc(plane_bytes));
        MDX_HIP(hipMemsetAsync(d_planes.ptr, 0, plane_bytes, stream));
        MDX_TRY(d_state.alloc(sizeof(double) * 2 * n_blocks * R));
        const double *g = nullptr, *h = nullptr;
        if (d_time_ptr != nullptr) {
          MDX_TRY(d_coef.alloc(sizeof(double) * 2 * padded));
          hipLaunchKernelGGL(trapezoid_coeff_kernel,
                             dim3(static_cast<unsigned>(ceil_div(padded, kThreads))),
                             dim3(kThreads), 0, stream, d_time_ptr, R, padded,
                             d_coef.as<double>(), d_coef.as<double>() + padded);
          MDX_HIP(hipGetLastError());
          g = d_coef.as<double>();
          h = g + padded;
        }
#define MDX_LAUNCH_STRIP(TYPE, KK, WAVES)                                                    \
  do {                                                                                       \
    const size_t lds_t = strip_lds_bytes(WAVES);                                             \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(acf_strip_kernel<TYPE, KK, WAVES>),                   \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_t)));               \
    hipLaunchKernelGGL((acf_strip_kernel<TYPE, KK, WAVES>),                                  \
                       dim3(static_cast<unsigned>(n_blocks)), dim3(WAVES * 64), lds_t,       \
                       stream, d_planes.as<TYPE>(), plane_pitch, k, t0, correlation_time,    \
                       win_begin + ws, nw, wpb, R, d_weight.as<double>(), g, h,              \
                       integral_scale, block_offset, d_acf.as<double>(),                     \
                       d_mean.as<double>(), d_m2.as<double>(), d_state.as<double>(),         \
                       d_state.as<double>() + n_blocks * R);                                 \
  } while (0)
#define MDX_LAUNCH_STRIP_T(TYPE)                                                             \
  do {                                                                                       \
    hipLaunchKernelGGL(series_planes_kernel<TYPE>,                                           \
                       dim3(static_cast<unsigned>(ceil_div(t_len * k, kThreads))),           \
                       dim3(kThreads), 0, stream, static_cast<const TYPE *>(dx), k, t0,      \
                       t_len, plane_pitch, d_planes.as<TYPE>());                             \
    MDX_HIP(hipGetLastError());                                                              \
    if (k == 3 && strip_waves == 8)                                                          \
      MDX_LAUNCH_STRIP(TYPE, 3, 8);                                                          \
    else if (k == 3 && strip_waves == 4)                                                     \
      MDX_LAUNCH_STRIP(TYPE, 3, 4);                                                          \
    else if (k == 3)                                                                         \
      MDX_LAUNCH_STRIP(TYPE, 3, 2);                                                          \
    else if (strip_waves == 8)                                                               \
      MDX_LAUNCH_STRIP(TYPE, 0, 8);                                                          \
    else if (strip_waves == 4)                                                               \
      MDX_LAUNCH_STRIP(TYPE, 0, 4);                                                          \
    else                                                                                     \
      MDX_LAUNCH_STRIP(TYPE, 0, 2);                                                          \
  } while (0)
        if (dtype == MDX_F32)
          MDX_LAUNCH_STRIP_T(float);
        else
          MDX_LAUNCH_STRIP_T(double);
#undef MDX_LAUNCH_STRIP_T
#undef MDX_LAUNCH_STRIP
      } else if (cross) {
        const double *table = band ? d_C.as<double>() : nullptr;
        if (dtype == MDX_F32) {
          using P = DistinctProduct<float>;
          P prod{static_cast<const float *>(job.cross_a), static_cast<const float *>(job.cross_b),
                 k, job.cross_weight, table, t0, R, job.self_weight};
          MDX_LAUNCH_SLIDE(P, prod);
        } else {
          using P = DistinctProduct<double>;
          P prod{static_cast<const double *>(job.cross_a),
                 static_cast<const double *>(job.cross_b), k, job.cross_weight, table, t0, R,
                 job.self_weight};
          MDX_LAUNCH_SLIDE(P, prod);
        }
      } else if (band) {
        BandProduct prod{d_C.as<double>(), t0, R};
        MDX_LAUNCH_SLIDE(BandProduct, prod);
      } else if (dtype == MDX_F32) {
        MDX_LAUNCH_SLIDE_SERIES(float);
      } else {
        MDX_LAUNCH_SLIDE_SERIES(double);
      }
#undef MDX_LAUNCH_SLIDE_SERIES
#undef MDX_LAUNCH_SLIDE_B
#undef MDX_LAUNCH_SLIDE
      MDX_HIP(hipGetLastError());
      block_offset += n_blocks;
    }
  }
  return MDX_OK;
}

// Merges the per-block results on the device (out = [acf R | mean R-1 | M2 R-1]) and
// adds / stores them into the caller's arrays.
int merge_blocks(hipStream_t stream, DeviceBuffer &d_acf, DeviceBuffer &d_mean,
                 DeviceBuffer &d_m2, const std::vector<int64_t> &block_windows, int R,
                 bool merge_moments, double *acf_sum, double *integral_mean,
                 double *integral_m2) {
  const int64_t n_blocks_b = static_cast<int64_t>(block_windows.size());
  const int n_int = R > 1 ? R - 1 : 1;
  DeviceBuffer d_counts, d_out;
  MDX_TRY(d_counts.alloc(sizeof(int64_t) * n_blocks_b));
  MDX_HIP(hipMemcpyAsync(d_counts.ptr, block_windows.data(), sizeof(int64_t) * n_blocks_b,
                         hipMemcpyHostToDevice, stream));
  MDX_TRY(d_out.alloc(sizeof(double) * (static_cast<size_t>(R) + 2 * n_int)));
  hipLaunchKernelGGL(acf_merge_kernel,
                     dim3(static_cast<unsigned>(ceil_div(R, kThreads / 64))), dim3(kThreads),
                     0, stream, d_acf.as<double>(),
                     merge_moments ? d_mean.as<double>() : nullptr,
                     merge_moments ? d_m2.as<double>() : nullptr, d_counts.as<int64_t>(),
                     n_blocks_b, R, d_out.as<double>());
  MDX_HIP(hipGetLastError());
  std::vector<double> h_out(static_cast<size_t>(R) + 2 * n_int);
  MDX_HIP(hipMemcpyAsync(h_out.data(), d_out.ptr, sizeof(double) * h_out.size(),
                         hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  for (int m = 0; m < R; ++m) acf_sum[m] += h_out[m];
  if (merge_moments) {
    for (int i = 0; i < R - 1; ++i) {
      integral_mean[i] = h_out[R + i];
      integral_m2[i] = h_out[R + (R - 1) + i];
    }
  }
  return MDX_OK;
}

}  // namespace

extern "C" {

int mdx_acf_windows(const void *x, int dtype, int location, int64_t n_series,
                    int64_t n_frames, int k, int64_t correlation_time, int64_t win_begin,
                    int64_t win_end, const int32_t *tau, int n_tau, double scale,
                    double integral_scale, const double *time, double *acf_sum,
                    double *integral_mean, double *integral_m2, int64_t *n_windows) {
  TauInfo ti;
  MDX_TRY(check_windows(x, dtype, location, n_series, n_frames, k, correlation_time,
                        win_begin, win_end, tau, n_tau, &ti));
  MDX_REQUIRE(acf_sum != nullptr, "acf_sum is null");
  const bool want_integral = (time != nullptr);
  if (want_integral)
    MDX_REQUIRE(integral_mean != nullptr && integral_m2 != nullptr,
                "integral outputs are null");
  const int R = n_tau;
  const int64_t n_win = win_end - win_begin;
  if (n_windows) *n_windows = n_win;
  if (want_integral) {
    for (int i = 0; i + 1 < R; ++i) {
      integral_mean[i] = 0.0;
      integral_m2[i] = 0.0;
    }
  }
  if (n_win == 0) return MDX_OK;

  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  DeviceBuffer x_staged, d_tau, d_time, d_C, d_acf, d_mean, d_m2;
  const void *dx = nullptr;
  MDX_TRY(to_device(x, location,
                    static_cast<size_t>(n_series) * n_frames * k * dtype_size(dtype), x_staged,
                    stream, &dx));
  MDX_TRY(d_tau.alloc(sizeof(int32_t) * R));
  MDX_HIP(hipMemcpyAsync(d_tau.ptr, tau, sizeof(int32_t) * R, hipMemcpyHostToDevice, stream));
  if (want_integral) {
    MDX_TRY(d_time.alloc(sizeof(double) * R));
    MDX_HIP(hipMemcpyAsync(d_time.ptr, time, sizeof(double) * R, hipMemcpyHostToDevice, stream));
  }

  // per-block results of whichever path runs; block_windows[b] = windows of block b
  std::vector<int64_t> block_windows;
  const int n_int = R > 1 ? R - 1 : 1;
  const bool slide = ti.identity && R >= 2 && 4 * correlation_time <= R;
  if (slide) {
    SlideJob job;
    job.dtype = dtype;
    job.series = dx;
    job.n_series = n_series;
    job.n_frames = n_frames;
    job.k = k;
    job.ct = correlation_time;
    job.win_begin = win_begin;
    job.n_win = n_win;
    job.R = R;
    job.scale = scale;
    job.integral_scale = integral_scale;
    job.d_time = want_integral ? d_time.as<double>() : nullptr;
    job.band = n_series > 8;
    MDX_TRY(slide_windows(job, stream, d_C, d_acf, d_mean, d_m2, block_windows));
  } else {
    // stage A geometry
    const int n_pairs = (R + 1) / 2;
    int cap = (kMaxOwned * kThreads) / n_pairs;
    if (cap < 1) cap = 1;
    if (cap > 8) cap = 8;
    const int wpb = windows_per_block(correlation_time, ti.D, k, n_win, cap, 96 * 1024);
    const size_t lds_a =
        static_cast<size_t>((wpb - 1) * correlation_time + ti.D) * k * sizeof(double) +
        sizeof(int32_t) * R + 16;
    MDX_REQUIRE(lds_a <= 160 * 1024 - 512,
                "data_range %d with %d dims needs %zu bytes of LDS, more than a gfx950 "
                "compute unit has",
                ti.D, k, lds_a);
    MDX_REQUIRE(static_cast<int64_t>(wpb) * n_pairs <= kMaxOwned * kThreads,
                "n_tau too large for the direct kernel");
    const int64_t win_blocks = ceil_div(n_win, wpb);
    // series chunks: enough blocks to fill the chip, bounded partial-sum memory
    int64_t n_chunks = ceil_div(2048, win_blocks);
    if (n_chunks > n_series) n_chunks = n_series;
    if (n_chunks > 64) n_chunks = 64;
    const size_t c_bytes_one = sizeof(double) * static_cast<size_t>(n_win) * R;
    while (n_chunks > 1 && c_bytes_one * n_chunks > (size_t(8) << 30)) --n_chunks;
    const int64_t series_per_chunk = ceil_div(n_series, n_chunks);
    n_chunks = ceil_div(n_series, series_per_chunk);
    MDX_REQUIRE(win_blocks <= 2147483647LL && n_chunks <= 65535, "grid too large");
    MDX_TRY(d_C.alloc(c_bytes_one * n_chunks));

    dim3 grid_a(static_cast<unsigned>(win_blocks), static_cast<unsigned>(n_chunks));
#define MDX_LAUNCH_ACF(TYPE, KK)                                                          \
    do {                                                                                    \
      MDX_HIP(hipFuncSetAttribute(                                                          \
          reinterpret_cast<const void *>(acf_partial_kernel<TYPE, KK>),                     \
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_a)));            \
      hipLaunchKernelGGL((acf_partial_kernel<TYPE, KK>), grid_a, dim3(kThreads), lds_a,     \
                         stream, static_cast<const TYPE *>(dx), n_series, n_frames, k,      \
                         correlation_time, win_begin, n_win, wpb, d_tau.as<int32_t>(), R,   \
                         ti.D, ti.identity ? 1 : 0, series_per_chunk, d_C.as<double>());    \
    } while (0)
    {
      KernelTimer timer(MDX_SLOT_ACF_PARTIAL, stream);
      if (dtype == MDX_F32) {
        if (k == 3) MDX_LAUNCH_ACF(float, 3); else MDX_LAUNCH_ACF(float, 0);
      } else {
        if (k == 3) MDX_LAUNCH_ACF(double, 3); else MDX_LAUNCH_ACF(double, 0);
      }
    }
#undef MDX_LAUNCH_ACF
    MDX_HIP(hipGetLastError());

    // stage B geometry: contiguous window ranges per block
    int64_t n_blocks_b = n_win < 1024 ? n_win : 1024;
    const int64_t wins_per_block_b = ceil_div(n_win, n_blocks_b);
    n_blocks_b = ceil_div(n_win, wins_per_block_b);
    MDX_TRY(d_acf.alloc(sizeof(double) * n_blocks_b * R));
    MDX_TRY(d_mean.alloc(sizeof(double) * n_blocks_b * n_int));
    MDX_TRY(d_m2.alloc(sizeof(double) * n_blocks_b * n_int));
    const size_t lds_b = sizeof(double) * (static_cast<size_t>(R) + kThreads);
    MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(acf_finish_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds_b)));
    {
      KernelTimer timer(MDX_SLOT_ACF_FINISH, stream);
      hipLaunchKernelGGL(acf_finish_kernel, dim3(static_cast<unsigned>(n_blocks_b)),
                         dim3(kThreads), lds_b, stream, d_C.as<double>(),
                         static_cast<int>(n_chunks), n_win, wins_per_block_b, R, scale,
                         integral_scale, want_integral ? d_time.as<double>() : nullptr,
                         d_acf.as<double>(), d_mean.as<double>(), d_m2.as<double>());
    }
    MDX_HIP(hipGetLastError());

    for (int64_t b = 0; b < n_blocks_b; ++b)
      block_windows.push_back(std::min(wins_per_block_b, n_win - b * wins_per_block_b));
  }
  return merge_blocks(stream, d_acf, d_mean, d_m2, block_windows, R, want_integral && R > 1,
                      acf_sum, integral_mean, integral_m2);
}

int mdx_ccf_windows(const void *a, const void *b, const void *self_atoms, int dtype,
                    int location, int64_t n_self_atoms, int64_t n_frames, int k,
                    int64_t correlation_time, int64_t win_begin, int64_t win_end,
                    int data_range, double cross_weight, double self_weight,
                    const double *time, double *cf_sum, double *integral_mean,
                    double *integral_m2, int64_t *n_windows) {
  MDX_REQUIRE(a != nullptr && b != nullptr && cf_sum != nullptr, "null argument");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  MDX_REQUIRE(n_frames >= 1 && k >= 1 && k <= 16, "bad series shape");
  MDX_REQUIRE(correlation_time >= 1, "correlation_time must be >= 1");
  MDX_REQUIRE(data_range >= 2 && data_range <= 8192, "data_range out of range (2..8192)");
  MDX_REQUIRE(win_begin >= 0 && win_end >= win_begin, "bad window range");
  MDX_REQUIRE(self_atoms == nullptr || n_self_atoms >= 1, "empty self term");
  const int R = data_range;
  const int64_t n_win = win_end - win_begin;
  if (n_win > 0) {
    const int64_t last = (win_end - 1) * correlation_time + R;
    MDX_REQUIRE(last <= n_frames,
                "window %lld needs frames up to %lld but the batch holds %lld",
                (long long)(win_end - 1), (long long)last, (long long)n_frames);
  }
  const bool want_integral = (time != nullptr);
  if (want_integral) {
    MDX_REQUIRE(integral_mean != nullptr && integral_m2 != nullptr,
                "integral outputs are null");
    for (int i = 0; i + 1 < R; ++i) {
      integral_mean[i] = 0.0;
      integral_m2[i] = 0.0;
    }
  }
  if (n_windows) *n_windows = n_win;
  if (n_win == 0) return MDX_OK;

  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  DeviceBuffer a_staged, b_staged, x_staged, d_time, d_C, d_acf, d_mean, d_m2;
  const size_t series_bytes = static_cast<size_t>(n_frames) * k * dtype_size(dtype);
  SlideJob job;
  MDX_TRY(to_device(a, location, series_bytes, a_staged, stream, &job.cross_a));
  if (b == a) {
    job.cross_b = job.cross_a;
  } else {
    MDX_TRY(to_device(b, location, series_bytes, b_staged, stream, &job.cross_b));
  }
  if (self_atoms != nullptr) {
    MDX_TRY(to_device(self_atoms, location, series_bytes * n_self_atoms, x_staged, stream,
                      &job.series));
    job.n_series = n_self_atoms;
    job.band = true;
  }
  if (want_integral) {
    MDX_TRY(d_time.alloc(sizeof(double) * R));
    MDX_HIP(hipMemcpyAsync(d_time.ptr, time, sizeof(double) * R, hipMemcpyHostToDevice, stream));
    job.d_time = d_time.as<double>();
  }
  job.dtype = dtype;
  job.n_frames = n_frames;
  job.k = k;
  job.ct = correlation_time;
  job.win_begin = win_begin;
  job.n_win = n_win;
  job.R = R;
  job.lag_normalise = false;
  job.cross_weight = cross_weight;
  job.self_weight = self_weight;
  std::vector<int64_t> block_windows;
  MDX_TRY(slide_windows(job, stream, d_C, d_acf, d_mean, d_m2, block_windows));
  return merge_blocks(stream, d_acf, d_mean, d_m2, block_windows, R, want_integral, cf_sum,
                      integral_mean, integral_m2);
}

int mdx_msd_windows(const void *x, int dtype, int location, int64_t n_series,
                    int64_t n_frames, int k, int64_t correlation_time, int64_t win_begin,
                    int64_t win_end, const int32_t *tau, int n_tau, double scale,
                    double *msd_sum) {
  TauInfo ti;
  MDX_TRY(check_windows(x, dtype, location, n_series, n_frames, k, correlation_time,
                        win_begin, win_end, tau, n_tau, &ti));
  MDX_REQUIRE(msd_sum != nullptr, "msd_sum is null");
  const int R = n_tau;
  const int64_t n_win = win_end - win_begin;
  if (n_win == 0) return MDX_OK;

  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  DeviceBuffer x_staged, d_tau, d_partial, d_out;
  const void *dx = nullptr;
  MDX_TRY(to_device(x, location,
                    static_cast<size_t>(n_series) * n_frames * k * dtype_size(dtype), x_staged,
                    stream, &dx));
  MDX_TRY(d_tau.alloc(sizeof(int32_t) * R));
  MDX_HIP(hipMemcpyAsync(d_tau.ptr, tau, sizeof(int32_t) * R, hipMemcpyHostToDevice, stream));

  // Many series, consecutive origins and every lag: the band kernel evaluates the
  // squared displacements of a tile of origins x lags from one staged image per series.
  if (ti.identity && correlation_time == 1 && n_series >= 8 && R >= 16) {
    const bool narrow = R <= 128;
    // three components go to the matrix cores unless MDX_MSD_BAND_VALU is set; their
    // lag tile is 112 (covers the default data_range of 100 with 8 tiles of 16), 128 or 256
    const bool mfma = (k == 3) && getenv("MDX_MSD_BAND_VALU") == nullptr;
    // the 4x4x4 instruction unless MDX_BAND_MFMA16 asks for the 16x16x4 kernel
    // (its float32 staging addresses a series through a 32-bit buffer offset)
    const bool mfma4 = mfma && getenv("MDX_BAND_MFMA16") == nullptr &&
                       (dtype != MDX_F32 || n_frames * 12 < (int64_t(1) << 31));
    const int mfma_m = mfma4 ? (R <= Band4<27>::kM ? Band4<27>::kM : Band4<33>::kM)
                             : (R <= 112 ? 112 : (R <= 128 ? 128 : 256));
    const int tile_t = mfma4 ? Band4<33>::kT
                             : (mfma ? 64 : (narrow ? BandTile<16>::T : BandTile<32>::T));
    const int tile_m = mfma ? mfma_m : (narrow ? BandTile<16>::M : BandTile<32>::M);
    const size_t elem = dtype_size(dtype);
    const size_t lds_q =
        mfma4 ? (mfma_m == Band4<27>::kM ? Band4<27>::lds_bytes(true)
                                         : Band4<33>::lds_bytes(true))
        : mfma ? (mfma_m == 112 ? MsdMf<112>::lds_bytes(elem)
                                : (mfma_m == 128 ? MsdMf<128>::lds_bytes(elem)
                                                 : MsdMf<256>::lds_bytes(elem)))
               : (narrow ? BandTile<16>::lds_bytes(k) : BandTile<32>::lds_bytes(k));
    const int64_t tiles_t = ceil_div(n_win, tile_t);
    const int64_t tiles_m = ceil_div(R, tile_m);
    // enough blocks that the last, partly filled round of resident blocks is a small share
    // enough blocks to fill the chip several times over, the last round of resident blocks
    // nearly full, and blocks long enough that their start and the final sums stay small
    int64_t n_chunks;
    if (mfma) {
      const int64_t most = n_series / 8 < 64 ? n_series / 8 : 64;
      const int64_t base = ceil_div(4096, tiles_t * tiles_m);
      n_chunks = band_chunks(tiles_t * tiles_m, base < most ? base : most, most);
    } else {
      n_chunks = ceil_div(2048, tiles_t * tiles_m);
      if (n_chunks > n_series / 8) n_chunks = n_series / 8;
    }
    if (n_chunks > 65535) n_chunks = 65535;
    if (n_chunks < 1) n_chunks = 1;
    int64_t series_per_chunk = ceil_div(n_series, n_chunks);
    if (mfma) series_per_chunk = 8 * ceil_div(series_per_chunk, 8);  // whole staging rounds
    n_chunks = ceil_div(n_series, series_per_chunk);
    MDX_REQUIRE(tiles_t <= 2147483647LL, "grid too large");
    const int64_t n_blocks = tiles_t * n_chunks;
    MDX_TRY(d_partial.alloc(sizeof(double) * n_blocks * R));
    MDX_TRY(d_out.alloc(sizeof(double) * R));
    dim3 grid_q(band_grid(tiles_t, tiles_m, n_chunks));
    {
      KernelTimer timer(MDX_SLOT_MSD, stream);
#define MDX_LAUNCH_MSD_BAND(TYPE, KK, MGG)                                                   \
  do {                                                                                       \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(acf_band_kernel<TYPE, KK, MGG, true>),                \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_q)));               \
    hipLaunchKernelGGL((acf_band_kernel<TYPE, KK, MGG, true>), grid_q, dim3(kThreads), lds_q, \
                       stream, static_cast<const TYPE *>(dx), n_series, n_frames, k,         \
                       win_begin, n_win, R, series_per_chunk, d_partial.as<double>(),        \
                       tiles_t, static_cast<int>(tiles_m), static_cast<int>(n_chunks));      \
  } while (0)
#define MDX_LAUNCH_MSD_BAND_K(TYPE, MGG)                                                     \
  do {                                                                                       \
    if (k == 3) MDX_LAUNCH_MSD_BAND(TYPE, 3, MGG); else MDX_LAUNCH_MSD_BAND(TYPE, 0, MGG);   \
  } while (0)
#define MDX_LAUNCH_MSD_MFMA(TYPE, MM)                                                         \
  do {                                                                                       \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(msd_band_mfma_kernel<TYPE, MM>),                      \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_q)));               \
    hipLaunchKernelGGL((msd_band_mfma_kernel<TYPE, MM>), grid_q, dim3(kThreads), lds_q,      \
                       stream, static_cast<const TYPE *>(dx), n_series, n_frames, win_begin, \
                       n_win, R, series_per_chunk, d_partial.as<double>(), tiles_t,          \
                       static_cast<int>(tiles_m), static_cast<int>(n_chunks));               \
  } while (0)
#define MDX_LAUNCH_MSD_BAND4(TYPE, NQQ)                                                       \
  do {                                                                                       \
    MDX_HIP(hipFuncSetAttribute(                                                             \
        reinterpret_cast<const void *>(band4_kernel<TYPE, NQQ, true>),                       \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_q)));               \
    hipLaunchKernelGGL((band4_kernel<TYPE, NQQ, true>), grid_q, dim3(kB4Threads), lds_q,       \
                       stream, static_cast<const TYPE *>(dx), n_series, n_frames, win_begin, \
                       n_win, R, series_per_chunk, d_partial.as<double>(), tiles_t,          \
                       static_cast<int>(tiles_m), static_cast<int>(n_chunks));               \
  } while (0)
      if (mfma4) {
        if (dtype == MDX_F32) {
          if (mfma_m == Band4<27>::kM) MDX_LAUNCH_MSD_BAND4(float, 27);
          else MDX_LAUNCH_MSD_BAND4(float, 33);
        } else {
          if (mfma_m == Band4<27>::kM) MDX_LAUNCH_MSD_BAND4(double, 27);
          else MDX_LAUNCH_MSD_BAND4(double, 33);
        }
      } else if (mfma) {
        if (dtype == MDX_F32) {
          if (mfma_m == 112) MDX_LAUNCH_MSD_MFMA(float, 112);
          else if (mfma_m == 128) MDX_LAUNCH_MSD_MFMA(float, 128);
          else MDX_LAUNCH_MSD_MFMA(float, 256);
        } else {
          if (mfma_m == 112) MDX_LAUNCH_MSD_MFMA(double, 112);
          else if (mfma_m == 128) MDX_LAUNCH_MSD_MFMA(double, 128);
          else MDX_LAUNCH_MSD_MFMA(double, 256);
        }
      } else if (dtype == MDX_F32) {
        if (narrow) MDX_LAUNCH_MSD_BAND_K(float, 16); else MDX_LAUNCH_MSD_BAND_K(float, 32);
      } else {
        if (narrow) MDX_LAUNCH_MSD_BAND_K(double, 16); else MDX_LAUNCH_MSD_BAND_K(double, 32);
      }
#undef MDX_LAUNCH_MSD_BAND4
#undef MDX_LAUNCH_MSD_MFMA
#undef MDX_LAUNCH_MSD_BAND_K
#undef MDX_LAUNCH_MSD_BAND
    }
    MDX_HIP(hipGetLastError());
    hipLaunchKernelGGL(combine_blocks_kernel,
                       dim3(static_cast<unsigned>(ceil_div(R, kThreads / 64))), dim3(kThreads), 0,
                       stream, d_partial.as<double>(), n_blocks, R, scale, d_out.as<double>());
    MDX_HIP(hipGetLastError());
    std::vector<double> h(static_cast<size_t>(R));
    MDX_HIP(hipMemcpyAsync(h.data(), d_out.ptr, sizeof(double) * R, hipMemcpyDeviceToHost,
                           stream));
    MDX_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < R; ++i) msd_sum[i] += h[i];
    return MDX_OK;
  }

  const int wpb = windows_per_block(correlation_time, ti.D, k, n_win, 256, 48 * 1024);
  const size_t lds =
      static_cast<size_t>((wpb - 1) * correlation_time + ti.D) * k * sizeof(double) + 16;
  MDX_REQUIRE(lds <= 160 * 1024 - 512, "data_range %d needs %zu bytes of LDS", ti.D, lds);
  const int64_t win_blocks = ceil_div(n_win, wpb);
  int64_t n_chunks = ceil_div(4096, win_blocks);
  if (n_chunks > n_series) n_chunks = n_series;
  if (n_chunks > 65535) n_chunks = 65535;
  const int64_t series_per_chunk = ceil_div(n_series, n_chunks);
  n_chunks = ceil_div(n_series, series_per_chunk);
  MDX_REQUIRE(win_blocks <= 2147483647LL, "grid too large");
  const int64_t n_blocks = win_blocks * n_chunks;
  MDX_TRY(d_partial.alloc(sizeof(double) * n_blocks * R));
  MDX_TRY(d_out.alloc(sizeof(double) * R));
  dim3 grid(static_cast<unsigned>(win_blocks), static_cast<unsigned>(n_chunks));
  KernelTimer timer(MDX_SLOT_MSD, stream);
  if (dtype == MDX_F32) {
    MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_kernel<float>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds)));
    hipLaunchKernelGGL(msd_kernel<float>, grid, dim3(kThreads), lds, stream,
                       static_cast<const float *>(dx), n_series, n_frames, k,
                       correlation_time, win_begin, n_win, wpb, d_tau.as<int32_t>(), R, ti.D,
                       series_per_chunk, d_partial.as<double>());
  } else {
    MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds)));
    hipLaunchKernelGGL(msd_kernel<double>, grid, dim3(kThreads), lds, stream,
                       static_cast<const double *>(dx), n_series, n_frames, k,
                       correlation_time, win_begin, n_win, wpb, d_tau.as<int32_t>(), R, ti.D,
                       series_per_chunk, d_partial.as<double>());
  }
  MDX_HIP(hipGetLastError());
  hipLaunchKernelGGL(combine_blocks_kernel, dim3(static_cast<unsigned>(ceil_div(R, kThreads / 64))),
                     dim3(kThreads), 0, stream, d_partial.as<double>(), n_blocks, R, scale,
                     d_out.as<double>());
  MDX_HIP(hipGetLastError());
  std::vector<double> h(static_cast<size_t>(R));
  MDX_HIP(hipMemcpyAsync(h.data(), d_out.ptr, sizeof(double) * R, hipMemcpyDeviceToHost,
                         stream));
  MDX_HIP(hipStreamSynchronize(stream));
  for (int i = 0; i < R; ++i) msd_sum[i] += h[i];
  return MDX_OK;
}

}  // extern "C"
