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

// floor(fl(v / box)) and rint(fl(v / box)) without the divide in the common case:
// the product with the rounded reciprocal is within 2^-51 (relative) of the
// correctly rounded quotient, so both round to the same integer unless the product
// lies that close to an integer (floor) or to a half-integer (rint); only then the
// IEEE divide decides.
__device__ __forceinline__ double floor_quotient(double v, double box, double inv_box) {
  const double q = v * inv_box;
  const double near = fabs(q - rint(q));
  if (__builtin_expect(near <= fabs(q) * 0x1p-50 + 0x1p-1000, 0)) return floor(v / box);
  return floor(q);
}
__device__ __forceinline__ double rint_quotient(double v, double box, double inv_box) {
  const double q = v * inv_box;
  const double n = rint(q);
  const double near = fabs(fabs(q - n) - 0.5);
  if (__builtin_expect(near <= fabs(q) * 0x1p-50 + 0x1p-1000, 0)) return rint(v / box);
  return n;
}

// ------------------------------------------------------------------ unwrap

// One block per atom.  jump[t] = rint((pos[t] - pos[t-1]) / box) (half to even, as
// tf.math.round); image[t] = last_image - sum_{u <= t} jump[u]; out = pos + image * box.
// The jumps are integers, so their float64 prefix sums are exact in any order and
// the block scan reproduces the sequential cumsum bit for bit.
constexpr int kPerThread = 4;  // consecutive frames (unwrap) / elements (maps) per thread

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void unwrap_kernel(
    const TIn *__restrict__ pos, int64_t n_frames, Box3 box, int have_carry,
    double *__restrict__ last_pos, double *__restrict__ last_image, TOut *__restrict__ out) {
  // A chunk of kThreads * kPerThread frames goes through LDS both ways, so global
  // accesses are contiguous while a thread walks over kPerThread consecutive frames.
  constexpr int kChunk = kThreads * kPerThread;
  __shared__ double wave_tot[3][kThreads / 64];
  __shared__ TIn in_s[3 + 3 * kChunk];  // [0..3): the frame before the chunk
  __shared__ TOut out_s[3 * kChunk];
  const int64_t atom = blockIdx.x;
  const TIn *x = pos + atom * n_frames * 3;
  TOut *y = out + atom * n_frames * 3;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double base[3], inv_box[3];
  for (int d = 0; d < 3; ++d) {
    inv_box[d] = 1.0 / box.v[d];
    base[d] = have_carry ? last_image[atom * 3 + d] : 0.0;  // image before this chunk
  }
  for (int64_t t0 = 0; t0 < n_frames; t0 += kChunk) {
    const int64_t n_here = min(static_cast<int64_t>(kChunk), n_frames - t0);
    __syncthreads();  // the previous chunk's readers of in_s / out_s are done
    for (int e = threadIdx.x; e < 3 * n_here; e += kThreads) in_s[3 + e] = x[t0 * 3 + e];
    if (threadIdx.x < 3) {
      if (t0 > 0)
        in_s[threadIdx.x] = x[(t0 - 1) * 3 + threadIdx.x];
      else if (!have_carry)
        in_s[threadIdx.x] = x[threadIdx.x];
    }
    __syncthreads();
    const int fb = threadIdx.x * kPerThread;  // first frame of this thread in the chunk
    double cur[kPerThread][3], cum[kPerThread][3], total[3];
    for (int d = 0; d < 3; ++d) {
      double prev = 0.0;
      if (fb < n_here) {
        // with a carry the frame before the very first one comes as float64
        prev = (t0 == 0 && fb == 0 && have_carry) ? last_pos[atom * 3 + d]
                                                  : static_cast<double>(in_s[3 * fb + d]);
      }
      double run = 0.0;
#pragma unroll
      for (int f = 0; f < kPerThread; ++f) {
        const bool valid = fb + f < n_here;
        cur[f][d] = valid ? static_cast<double>(in_s[3 + 3 * (fb + f) + d]) : 0.0;
        if (valid) run = run + rint_quotient(cur[f][d] - prev, box.v[d], inv_box[d]);
        cum[f][d] = run;
        prev = cur[f][d];
      }
      total[d] = run;
    }
    double incl[3];
    for (int d = 0; d < 3; ++d) incl[d] = total[d];
    for (int off = 1; off < 64; off <<= 1) {
      for (int d = 0; d < 3; ++d) {
        const double up = __shfl_up(incl[d], off);
        if (lane >= off) incl[d] = incl[d] + up;
      }
    }
    if (lane == 63)
      for (int d = 0; d < 3; ++d) wave_tot[d][wave] = incl[d];
    __syncthreads();
    for (int d = 0; d < 3; ++d) {
      double before = 0.0, all = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) {
        if (w < wave) before = before + wave_tot[d][w];
        all = all + wave_tot[d][w];
      }
      const double preceding = (incl[d] - total[d]) + before;  // jumps before this thread
#pragma unroll
      for (int f = 0; f < kPerThread; ++f) {
        if (fb + f < n_here) {
          const double image = -(cum[f][d] + preceding) + base[d];
          out_s[3 * (fb + f) + d] = static_cast<TOut>(cur[f][d] + image * box.v[d]);
          if (t0 + fb + f == n_frames - 1) {
            last_pos[atom * 3 + d] = cur[f][d];
            last_image[atom * 3 + d] = image;
          }
        }
      }
      base[d] = base[d] - all;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * n_here; e += kThreads) y[t0 * 3 + e] = out_s[e];
  }
}

// --------------------------------------------------------- element-wise maps

// A thread owns kPerThread consecutive elements (16-byte accesses for float32).
template <int OP, typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void map_kernel(const TIn *__restrict__ x,
                                                       const TIn *__restrict__ aux,
                                                       int64_t n_elements, int64_t n_frames,
                                                       Box3 par, Box3 inv,
                                                       TOut *__restrict__ out) {
  const int64_t e0 =
      (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * kPerThread;
  if (e0 >= n_elements) return;
  const bool full = e0 + kPerThread <= n_elements;
  TIn in[kPerThread], ax[kPerThread];
  if (full) {
    struct alignas(sizeof(TIn) * kPerThread) Pack { TIn v[kPerThread]; };
    const Pack p = *reinterpret_cast<const Pack *>(x + e0);
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) in[j] = p.v[j];
    if (OP == MDX_MAP_UNWRAP_INDICES) {
      const Pack q = *reinterpret_cast<const Pack *>(aux + e0);
#pragma unroll
      for (int j = 0; j < kPerThread; ++j) ax[j] = q.v[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      in[j] = (e0 + j < n_elements) ? x[e0 + j] : TIn(0);
      if (OP == MDX_MAP_UNWRAP_INDICES) ax[j] = (e0 + j < n_elements) ? aux[e0 + j] : TIn(0);
    }
  }
  int d = static_cast<int>(e0 % 3);
  int64_t t = (OP == MDX_MAP_VELOCITY) ? (e0 / 3) % n_frames : 0;
  TOut res[kPerThread];
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const double v = static_cast<double>(in[j]);
    double r;
    if (OP == MDX_MAP_WRAP) {
      r = v - floor_quotient(v, par.v[d], inv.v[d]) * par.v[d];
    } else if (OP == MDX_MAP_WRAP_CENTERED) {
      const double half = par.v[d] / 2.0;
      const double u = v + half;
      r = (u - floor_quotient(u, par.v[d], inv.v[d]) * par.v[d]) - half;
    } else if (OP == MDX_MAP_SCALE) {
      r = v * par.v[d];
    } else if (OP == MDX_MAP_UNWRAP_INDICES) {
      r = v + static_cast<double>(ax[j]) * par.v[d];
    } else {  // MDX_MAP_VELOCITY: forward difference, the last frame repeats the one before
      const int64_t e = e0 + j;
      if (e < n_elements) {
        if (t + 1 < n_frames) {
          const double next = (j + 3 < kPerThread && full) ? static_cast<double>(in[j + 3])
                                                           : static_cast<double>(x[e + 3]);
          r = (next - v) / par.v[0];
        } else {
          r = (v - static_cast<double>(x[e - 3])) / par.v[0];
        }
      } else {
        r = 0.0;
      }
    }
    res[j] = static_cast<TOut>(r);
    if (++d == 3) {
      d = 0;
      if (OP == MDX_MAP_VELOCITY && ++t == n_frames) t = 0;
    }
  }
  if (full) {
    struct alignas(sizeof(TOut) * kPerThread) PackOut { TOut v[kPerThread]; };
    PackOut p;
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) p.v[j] = res[j];
    *reinterpret_cast<PackOut *>(out + e0) = p;
  } else {
#pragma unroll
    for (int j = 0; j < kPerThread; ++j)
      if (e0 + j < n_elements) out[e0 + j] = res[j];
  }
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
  MDX_REQUIRE(reinterpret_cast<uintptr_t>(dx) % 32 == 0 &&
                  reinterpret_cast<uintptr_t>(dout) % 32 == 0 &&
                  reinterpret_cast<uintptr_t>(daux) % 32 == 0,
              "device arrays must be 32-byte aligned");
  Box3 par{{params[0], params[1], params[2]}};
  Box3 inv{{1.0 / params[0], 1.0 / params[1], 1.0 / params[2]}};
  const int64_t n_blocks = ceil_div(static_cast<int64_t>(n), kThreads * kPerThread);
  MDX_REQUIRE(n_blocks <= 2147483647LL, "array too large");
  const dim3 grid(static_cast<unsigned>(n_blocks));
#define MDX_LAUNCH_MAP_T(OP, TIN, TOUT)                                                    \
  hipLaunchKernelGGL((map_kernel<OP, TIN, TOUT>), grid, dim3(kThreads), 0, stream,         \
                     static_cast<const TIN *>(dx), static_cast<const TIN *>(daux),         \
                     static_cast<int64_t>(n), n_frames, par, inv, static_cast<TOUT *>(dout))
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
