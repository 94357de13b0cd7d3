// Per-atom coordinate transformations on [atoms][frames][3] arrays: unwrapping by
// jump detection (a scan along the frame axis), wrapping, scaling, unwrapping by
// stored image indices and finite-difference velocities.
//
// Every one is a single pass over the data, bound by HBM bandwidth (one read and
// one write per element).  The arithmetic is float64 in the operation order of
// the reference's tensor expressions (separate multiply and add: the file is
// compiled with -ffp-contract=off), so results are bit-identical to a float64
// NumPy evaluation; a float32 output rounds once at the store, like the
// reference's float32 datasets.
#include "mdx_common.h"

using namespace mdx;

namespace {

constexpr int kThreads = 256;

struct Box3 {
  double v[3];
};

template <typename T>
__device__ __forceinline__ void store_value(T *dst, double v) {
  *dst = static_cast<T>(v);
}

// ------------------------------------------------------------------ unwrap

// One block per atom.  jump[t] = rint((pos[t] - pos[t-1]) / box) (half to even, as
// tf.math.round); image[t] = last_image - sum_{u <= t} jump[u]; out = pos + image * box.
// The jumps are integers, so their float64 prefix sums are exact in any order and
// the block scan reproduces the sequential cumsum bit for bit.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void unwrap_kernel(
    const TIn *__restrict__ pos, int64_t n_frames, Box3 box, int have_carry,
    double *__restrict__ last_pos, double *__restrict__ last_image, TOut *__restrict__ out) {
  __shared__ double wave_tot[3][kThreads / 64];
  const int64_t atom = blockIdx.x;
  const TIn *x = pos + atom * n_frames * 3;
  TOut *y = out + atom * n_frames * 3;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double first_prev[3], base[3];
  for (int d = 0; d < 3; ++d) {
    first_prev[d] = have_carry ? last_pos[atom * 3 + d] : static_cast<double>(x[d]);
    base[d] = have_carry ? last_image[atom * 3 + d] : 0.0;  // image before this chunk
  }
  for (int64_t t0 = 0; t0 < n_frames; t0 += kThreads) {
    const int64_t t = t0 + threadIdx.x;
    const bool valid = t < n_frames;
    double cur[3], incl[3];
    for (int d = 0; d < 3; ++d) {
      cur[d] = valid ? static_cast<double>(x[t * 3 + d]) : 0.0;
      const double prev =
          !valid ? 0.0 : (t == 0 ? first_prev[d] : static_cast<double>(x[(t - 1) * 3 + d]));
      const double diff = cur[d] - prev;
      incl[d] = valid ? rint(diff / box.v[d]) : 0.0;
    }
    for (int off = 1; off < 64; off <<= 1) {
      for (int d = 0; d < 3; ++d) {
        const double up = __shfl_up(incl[d], off);
        if (lane >= off) incl[d] = incl[d] + up;
      }
    }
    __syncthreads();  // the previous chunk's readers of wave_tot are done
    if (lane == 63)
      for (int d = 0; d < 3; ++d) wave_tot[d][wave] = incl[d];
    __syncthreads();
    double chunk_total[3];
    for (int d = 0; d < 3; ++d) {
      double before = 0.0, all = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) {
        if (w < wave) before = before + wave_tot[d][w];
        all = all + wave_tot[d][w];
      }
      chunk_total[d] = all;
      const double image = -(incl[d] + before) + base[d];
      if (valid) {
        store_value(&y[t * 3 + d], cur[d] + image * box.v[d]);
        if (t == n_frames - 1) {
          last_pos[atom * 3 + d] = cur[d];
          last_image[atom * 3 + d] = image;
        }
      }
    }
    for (int d = 0; d < 3; ++d) base[d] = base[d] - chunk_total[d];
  }
}

// --------------------------------------------------------- element-wise maps

template <int OP, typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void map_kernel(const TIn *__restrict__ x,
                                                       const TIn *__restrict__ aux,
                                                       int64_t n_elements, int64_t n_frames,
                                                       Box3 par, TOut *__restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= n_elements) return;
  const int d = static_cast<int>(e % 3);
  const double v = static_cast<double>(x[e]);
  double r;
  if (OP == MDX_MAP_WRAP) {
    r = v - floor(v / par.v[d]) * par.v[d];
  } else if (OP == MDX_MAP_WRAP_CENTERED) {
    const double half = par.v[d] / 2.0;
    const double u = v + half;
    r = (u - floor(u / par.v[d]) * par.v[d]) - half;
  } else if (OP == MDX_MAP_SCALE) {
    r = v * par.v[d];
  } else if (OP == MDX_MAP_UNWRAP_INDICES) {
    r = v + static_cast<double>(aux[e]) * par.v[d];
  } else {  // MDX_MAP_VELOCITY: forward difference, the last frame repeats the one before
    const int64_t t = (e / 3) % n_frames;
    if (t + 1 < n_frames)
      r = (static_cast<double>(x[e + 3]) - v) / par.v[0];
    else
      r = (v - static_cast<double>(x[e - 3])) / par.v[0];
  }
  store_value(&out[e], r);
}

int check_common(const void *x, int dtype, int location, int64_t n_atoms, int64_t n_frames,
                 const void *out, int out_dtype, int out_location) {
  MDX_REQUIRE(x != nullptr && out != nullptr, "null data pointer");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(out_dtype == MDX_F32 || out_dtype == MDX_F64,
              "out_dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  MDX_REQUIRE(out_location == MDX_HOST || out_location == MDX_DEVICE, "bad out_location");
  MDX_REQUIRE(n_atoms >= 1 && n_frames >= 1, "empty input");
  MDX_REQUIRE(n_atoms <= 2147483647LL, "too many atoms");
  return MDX_OK;
}

// device destination for the result: the caller's buffer or a staged one
int out_buffer(void *out, int out_location, size_t bytes, DeviceBuffer &staging, void **dptr) {
  if (out_location == MDX_DEVICE) {
    *dptr = out;
    return MDX_OK;
  }
  MDX_TRY(staging.alloc(bytes));
  *dptr = staging.ptr;
  return MDX_OK;
}

}  // namespace

extern "C" {

int mdx_unwrap_coordinates(const void *pos, int dtype, int location, int64_t n_atoms,
                           int64_t n_frames, const double box[3], int have_carry,
                           double *last_pos, double *last_image, void *out, int out_dtype,
                           int out_location) {
  MDX_TRY(check_common(pos, dtype, location, n_atoms, n_frames, out, out_dtype, out_location));
  MDX_REQUIRE(box != nullptr && last_pos != nullptr && last_image != nullptr, "null argument");
  for (int d = 0; d < 3; ++d) MDX_REQUIRE(box[d] > 0.0, "box lengths must be positive");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const size_t n = static_cast<size_t>(n_atoms) * n_frames * 3;
  DeviceBuffer x_staged, out_staged, d_last_pos, d_last_image;
  const void *dx = nullptr;
  void *dout = nullptr;
  MDX_TRY(to_device(pos, location, n * dtype_size(dtype), x_staged, stream, &dx));
  MDX_TRY(out_buffer(out, out_location, n * dtype_size(out_dtype), out_staged, &dout));
  const size_t carry_bytes = sizeof(double) * 3 * static_cast<size_t>(n_atoms);
  MDX_TRY(d_last_pos.alloc(carry_bytes));
  MDX_TRY(d_last_image.alloc(carry_bytes));
  if (have_carry) {
    MDX_HIP(hipMemcpyAsync(d_last_pos.ptr, last_pos, carry_bytes, hipMemcpyHostToDevice, stream));
    MDX_HIP(hipMemcpyAsync(d_last_image.ptr, last_image, carry_bytes, hipMemcpyHostToDevice,
                           stream));
  }
  Box3 b{{box[0], box[1], box[2]}};
  const dim3 grid(static_cast<unsigned>(n_atoms));
#define MDX_LAUNCH_UNWRAP(TIN, TOUT)                                                       \
  hipLaunchKernelGGL((unwrap_kernel<TIN, TOUT>), grid, dim3(kThreads), 0, stream,          \
                     static_cast<const TIN *>(dx), n_frames, b, have_carry ? 1 : 0,        \
                     d_last_pos.as<double>(), d_last_image.as<double>(),                   \
                     static_cast<TOUT *>(dout))
  if (dtype == MDX_F32) {
    if (out_dtype == MDX_F32) MDX_LAUNCH_UNWRAP(float, float); else MDX_LAUNCH_UNWRAP(float, double);
  } else {
    if (out_dtype == MDX_F32) MDX_LAUNCH_UNWRAP(double, float); else MDX_LAUNCH_UNWRAP(double, double);
  }
#undef MDX_LAUNCH_UNWRAP
  MDX_HIP(hipGetLastError());
  MDX_HIP(hipMemcpyAsync(last_pos, d_last_pos.ptr, carry_bytes, hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipMemcpyAsync(last_image, d_last_image.ptr, carry_bytes, hipMemcpyDeviceToHost,
                         stream));
  if (out_location == MDX_HOST)
    MDX_HIP(hipMemcpyAsync(out, dout, n * dtype_size(out_dtype), hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

int mdx_map_coordinates(int op, const void *x, const void *aux, int dtype, int location,
                        int64_t n_atoms, int64_t n_frames, const double params[3], void *out,
                        int out_dtype, int out_location) {
  MDX_TRY(check_common(x, dtype, location, n_atoms, n_frames, out, out_dtype, out_location));
  MDX_REQUIRE(op >= MDX_MAP_WRAP && op <= MDX_MAP_VELOCITY, "unknown map operation %d", op);
  MDX_REQUIRE(params != nullptr, "params is null");
  if (op == MDX_MAP_UNWRAP_INDICES) MDX_REQUIRE(aux != nullptr, "image indices are null");
  if (op == MDX_MAP_VELOCITY) {
    MDX_REQUIRE(n_frames >= 2, "a finite-difference velocity needs two frames");
    MDX_REQUIRE(params[0] != 0.0, "time step must not be zero");
  } else if (op != MDX_MAP_SCALE) {
    for (int d = 0; d < 3; ++d) MDX_REQUIRE(params[d] > 0.0, "box lengths must be positive");
  }
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const size_t n = static_cast<size_t>(n_atoms) * n_frames * 3;
  DeviceBuffer x_staged, aux_staged, out_staged;
  const void *dx = nullptr, *daux = nullptr;
  void *dout = nullptr;
  MDX_TRY(to_device(x, location, n * dtype_size(dtype), x_staged, stream, &dx));
  if (op == MDX_MAP_UNWRAP_INDICES)
    MDX_TRY(to_device(aux, location, n * dtype_size(dtype), aux_staged, stream, &daux));
  MDX_TRY(out_buffer(out, out_location, n * dtype_size(out_dtype), out_staged, &dout));
  Box3 par{{params[0], params[1], params[2]}};
  const dim3 grid(static_cast<unsigned>(ceil_div(static_cast<int64_t>(n), kThreads)));
  MDX_REQUIRE(ceil_div(static_cast<int64_t>(n), kThreads) <= 2147483647LL, "array too large");
#define MDX_LAUNCH_MAP_T(OP, TIN, TOUT)                                                    \
  hipLaunchKernelGGL((map_kernel<OP, TIN, TOUT>), grid, dim3(kThreads), 0, stream,         \
                     static_cast<const TIN *>(dx), static_cast<const TIN *>(daux),         \
                     static_cast<int64_t>(n), n_frames, par, static_cast<TOUT *>(dout))
#define MDX_LAUNCH_MAP(OP)                                                                 \
  do {                                                                                     \
    if (dtype == MDX_F32) {                                                                \
      if (out_dtype == MDX_F32) MDX_LAUNCH_MAP_T(OP, float, float);                        \
      else MDX_LAUNCH_MAP_T(OP, float, double);                                            \
    } else {                                                                               \
      if (out_dtype == MDX_F32) MDX_LAUNCH_MAP_T(OP, double, float);                       \
      else MDX_LAUNCH_MAP_T(OP, double, double);                                           \
    }                                                                                      \
  } while (0)
  switch (op) {
    case MDX_MAP_WRAP: MDX_LAUNCH_MAP(MDX_MAP_WRAP); break;
    case MDX_MAP_WRAP_CENTERED: MDX_LAUNCH_MAP(MDX_MAP_WRAP_CENTERED); break;
    case MDX_MAP_SCALE: MDX_LAUNCH_MAP(MDX_MAP_SCALE); break;
    case MDX_MAP_UNWRAP_INDICES: MDX_LAUNCH_MAP(MDX_MAP_UNWRAP_INDICES); break;
    default: MDX_LAUNCH_MAP(MDX_MAP_VELOCITY); break;
  }
#undef MDX_LAUNCH_MAP
#undef MDX_LAUNCH_MAP_T
  MDX_HIP(hipGetLastError());
  if (out_location == MDX_HOST)
    MDX_HIP(hipMemcpyAsync(out, dout, n * dtype_size(out_dtype), hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

}  // extern "C"
