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
//   msd_kernel           per block partial sums of squared displacements
//   combine kernels      fixed-order sums of the per-block partials
//
// All accumulation is float64; float32 input is widened exactly.
#include <algorithm>
#include <cmath>

#include "mdx_common.h"

using namespace mdx;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxOwned = 16;  // (window, lag pair) items per thread in the ACF kernel

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
            for (int d = 0; d < k; ++d) sum = sum + a[d] * b[d];
          }
          acc1[o] += sum;
          if (m2 != m1) {
            sum = 0.0;
            for (int n = 0; n < R - m2; ++n) {
              const double *a = base + n * k;
              const double *b = base + (n + m2) * k;
              for (int d = 0; d < k; ++d) sum = sum + a[d] * b[d];
            }
            acc2[o] += sum;
          }
        } else {
          for (int n = 0; n < R - m1; ++n) {
            const double *a = base + tau_l[n] * k;
            const double *b = base + tau_l[n + m1] * k;
            for (int d = 0; d < k; ++d) sum = sum + a[d] * b[d];
          }
          acc1[o] += sum;
          if (m2 != m1) {
            sum = 0.0;
            for (int n = 0; n < R - m2; ++n) {
              const double *a = base + tau_l[n] * k;
              const double *b = base + tau_l[n + m2] * k;
              for (int d = 0; d < k; ++d) sum = sum + a[d] * b[d];
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
    tot[threadIdx.x] = run;
    __syncthreads();
    double offset = 0.0;
    for (int t = 0; t < static_cast<int>(threadIdx.x); ++t) offset = offset + tot[t];
    run = offset;
    const double count = static_cast<double>(w - w0 + 1);
    for (int i = i0; i < i1; ++i) {
      run = run + (time[i + 1] - time[i]) * (cs[i + 1] + cs[i]) / 2.0;
      const double delta = run - mean[i];
      const double mu = mean[i] + delta / count;
      mean[i] = mu;
      m2[i] = m2[i] + delta * (run - mu);
    }
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
          sum = sum + diff * diff;
        }
      }
      out[i] = out[i] + sum;
    }
  }
}

// out[i] = scale * sum_b partial[b][i], blocks in index order
__global__ __launch_bounds__(kThreads) void combine_blocks_kernel(
    const double *__restrict__ partial, int64_t n_blocks, int R, double scale,
    double *__restrict__ out) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= R) return;
  double s = 0.0;
  for (int64_t b = 0; b < n_blocks; ++b) s = s + partial[b * R + i];
  out[i] = s * scale;
}

// ----------------------------------------------------------------- host side

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
  const int n_int = R > 1 ? R - 1 : 1;
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

  std::vector<double> h_acf(static_cast<size_t>(n_blocks_b) * R);
  std::vector<double> h_mean, h_m2;
  MDX_HIP(hipMemcpyAsync(h_acf.data(), d_acf.ptr, sizeof(double) * h_acf.size(),
                         hipMemcpyDeviceToHost, stream));
  if (want_integral && R > 1) {
    h_mean.resize(static_cast<size_t>(n_blocks_b) * (R - 1));
    h_m2.resize(h_mean.size());
    MDX_HIP(hipMemcpyAsync(h_mean.data(), d_mean.ptr, sizeof(double) * h_mean.size(),
                           hipMemcpyDeviceToHost, stream));
    MDX_HIP(hipMemcpyAsync(h_m2.data(), d_m2.ptr, sizeof(double) * h_m2.size(),
                           hipMemcpyDeviceToHost, stream));
  }
  MDX_HIP(hipStreamSynchronize(stream));

  for (int64_t b = 0; b < n_blocks_b; ++b)
    for (int m = 0; m < R; ++m) acf_sum[m] += h_acf[b * R + m];
  if (want_integral && R > 1) {
    // merge the per-block (count, mean, M2) triples in block order
    for (int i = 0; i < R - 1; ++i) {
      double n_a = 0.0, mean_a = 0.0, m2_a = 0.0;
      for (int64_t b = 0; b < n_blocks_b; ++b) {
        const int64_t wb0 = b * wins_per_block_b;
        const int64_t wb1 = std::min(n_win, wb0 + wins_per_block_b);
        const double n_b = static_cast<double>(wb1 - wb0);
        if (n_b <= 0.0) continue;
        const double mean_b = h_mean[b * (R - 1) + i];
        const double m2_b = h_m2[b * (R - 1) + i];
        const double delta = mean_b - mean_a;
        const double n_ab = n_a + n_b;
        mean_a = mean_a + delta * (n_b / n_ab);
        m2_a = m2_a + m2_b + delta * delta * (n_a * n_b / n_ab);
        n_a = n_ab;
      }
      integral_mean[i] = mean_a;
      integral_m2[i] = m2_a;
    }
  }
  return MDX_OK;
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
  hipLaunchKernelGGL(combine_blocks_kernel, dim3(static_cast<unsigned>(ceil_div(R, kThreads))),
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
