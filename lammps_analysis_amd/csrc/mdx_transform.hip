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

// The same transformation as a stream: float32 in, float32 out, wave = one atom walking
// down its frame axis in steps of 256 frames (lane = 4 consecutive frames).  The step's
// 3 KiB arrive with lane-contiguous loads (16 bytes per lane where the rows allow it, one
// word otherwise), pass through a wave-private LDS image in which every lane's 12 words
// sit kStride words apart (12: the 16-byte reads of sixteen lanes hit disjoint banks; 13
// for the word-wise variant), and leave the same way.  Only lanes of one wave exchange
// data and a wave's LDS operations complete in order, so there is no barrier; the next
// step's loads are in flight while the current one is computed.  Most frames do not jump:
// a quotient below 0.49 in magnitude rounds to zero whatever its last bits, and only the
// lanes beyond that take the exact rounding.  With 16-byte units a step in which no word
// moved by 0.48 box lengths or more (tested in float32 on the words as they were loaded,
// a word's predecessor being three words back: same lane or the lane before) has no jump
// at all; its results are the input plus the image shift of the step's start, word by
// word, and it bypasses LDS and the scan.
// Eight frames per lane (strides 28 and 25) halve the steps and measure the same: more
// registers, fewer waves per SIMD.
constexpr int kUwFrames = 4;
constexpr int kUwWords = 3 * kUwFrames;       // its words in a step
constexpr int kUwStep = 64 * kUwFrames;       // frames of a wave's step
constexpr int kUwWaves = kThreads / 64;
typedef float uw_f4 __attribute__((ext_vector_type(4)));

template <int VEC>
struct UwUnit;
template <>
struct UwUnit<4> {
  typedef uw_f4 type;
  static __device__ __forceinline__ uw_f4 zero() { return uw_f4{0.f, 0.f, 0.f, 0.f}; }
};
template <>
struct UwUnit<1> {
  typedef float type;
  static __device__ __forceinline__ float zero() { return 0.f; }
};

__device__ __forceinline__ double rint_quotient_near_zero(double v, double box, double inv_box) {
  const double q = v * inv_box;
  // |q| < 0.49 and q within 2^-51 (relative) of the rounded quotient: that one lies
  // below one half, so it rounds to zero
  if (__builtin_expect(fabs(q) < 0.49, 1)) return 0.0;
  return rint_quotient(v, box, inv_box);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void unwrap_stream_kernel(
    const float *__restrict__ pos, int64_t n_atoms, int64_t n_frames, Box3 box,
    int have_carry, double *__restrict__ last_pos, double *__restrict__ last_image,
    float *__restrict__ out) {
  typedef typename UwUnit<VEC>::type Unit;
  static_assert(kUwFrames == 4 || kUwFrames == 8, "lane strides below are for 4 or 8 frames");
  constexpr int kStride = VEC == 4 ? (kUwFrames == 8 ? 28 : 12) : kUwWords + 1;
  constexpr int kUnits = kUwWords / VEC;  // units a lane moves per step
  __shared__ __align__(16) float image[kUwWaves][64 * kStride];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t atom = static_cast<int64_t>(blockIdx.x) * kUwWaves + wave;
  if (atom >= n_atoms) return;  // no block-wide synchronisation below
  const float *x = pos + atom * n_frames * 3;
  float *y = out + atom * n_frames * 3;
  float *mine = image[wave];
  const int64_t n_units = n_frames * 3 / VEC;  // VEC == 4 only with n_frames % 4 == 0

  // LDS word of the unit u = lane + 64 j of a step
  int slot[kUnits];
#pragma unroll
  for (int j = 0; j < kUnits; ++j) {
    const int g = (lane + 64 * j) * VEC;
    slot[j] = (g / kUwWords) * kStride + g % kUwWords;
  }

  double inv_box[3], base[3], before[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    inv_box[d] = 1.0 / box.v[d];
    base[d] = have_carry ? last_image[atom * 3 + d] : 0.0;  // image before this step
    // the frame before the first one: handed over in float64, or the first frame itself
    before[d] = have_carry ? last_pos[atom * 3 + d] : static_cast<double>(x[d]);
  }

  // Quiet path (16-byte units): component of the word i of unit lane + 64 j is
  // (4 (lane + 64 j) + i) mod 3 = (lane + j + i) mod 3 in every step (a step is a whole
  // number of frames), so a lane meets the components in a fixed rotation: entry
  // (j + i) mod 3 of its three float32 bounds on a move without a jump and of its three
  // shifts image * box of the step's start.
  const int rot = lane % 3;
  float quiet_bound[3];
  double shift[3];
  float tail[3] = {0.f, 0.f, 0.f};  // the last three words of the step before
  auto pick = [&](int k, double v0, double v1, double v2) {
    const int d = (rot + k) % 3;
    return d == 0 ? v0 : (d == 1 ? v1 : v2);
  };
  auto set_shift = [&]() {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      shift[k] = pick(k, base[0] * box.v[0], base[1] * box.v[1], base[2] * box.v[2]);
  };
  if constexpr (VEC == 4) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // 0.48 box, rounded down to float32: a float32 difference below it means a true
      // difference below 0.4801 box and a quotient below 0.49
      const double b = 0.48 * pick(k, box.v[0], box.v[1], box.v[2]);
      float fb32 = static_cast<float>(b);
      if (static_cast<double>(fb32) > b) fb32 = __uint_as_float(__float_as_uint(fb32) - 1u);
      quiet_bound[k] = fb32;
    }
    set_shift();
  }

  Unit next[kUnits];
  // Units past the row's end re-read its last unit and are replaced by zero where they
  // are used: no branch around a load.
  auto fetch = [&](int64_t t0) {
    const int64_t u0 = t0 * 3 / VEC;
#pragma unroll
    for (int j = 0; j < kUnits; ++j) {
      const int64_t u = min(u0 + lane + 64 * j, n_units - 1);
      next[j] = __builtin_nontemporal_load(reinterpret_cast<const Unit *>(x) + u);
    }
  };

  fetch(0);
  for (int64_t t0 = 0; t0 < n_frames; t0 += kUwStep) {
    const int64_t u0 = t0 * 3 / VEC;
    const int units_here = static_cast<int>(min<int64_t>(64 * kUnits, n_units - u0));
    Unit cur_units[kUnits];
#pragma unroll
    for (int j = 0; j < kUnits; ++j)
      cur_units[j] = (lane + 64 * j < units_here) ? next[j] : UwUnit<VEC>::zero();
    if (t0 + kUwStep < n_frames) fetch(t0 + kUwStep);
    const int last_local = static_cast<int>(min<int64_t>(n_frames - 1 - t0, kUwStep));
    if constexpr (VEC == 4) {
      // The first step compares its first frame with the carry (float64) or with itself:
      // with a carry it takes the exact path.
      bool quiet = !(t0 == 0 && have_carry);
      float low[3] = {tail[0], tail[1], tail[2]};  // words 1..3 of the unit before lane 0's
#pragma unroll
      for (int j = 0; j < kUnits; ++j) {
        const uw_f4 u = cur_units[j];
        // the three words before this unit: words 1..3 of the previous lane's unit
        float p1 = __shfl_up(u.y, 1), p2 = __shfl_up(u.z, 1), p3 = __shfl_up(u.w, 1);
        if (lane == 0) {
          p1 = low[0]; p2 = low[1]; p3 = low[2];
        }
        const bool live = lane + 64 * j < units_here;
        const bool first = t0 == 0 && j == 0 && lane == 0;  // frame 0 precedes itself
        quiet = quiet && (!live ||
                          ((first || (fabsf(u.x - p1) < quiet_bound[j % 3] &&
                                      fabsf(u.y - p2) < quiet_bound[(j + 1) % 3] &&
                                      fabsf(u.z - p3) < quiet_bound[(j + 2) % 3])) &&
                           fabsf(u.w - u.x) < quiet_bound[j % 3]));
        low[0] = __shfl(u.y, 63);
        low[1] = __shfl(u.z, 63);
        low[2] = __shfl(u.w, 63);
      }
      if (__all(quiet)) {
#pragma unroll
        for (int j = 0; j < kUnits; ++j) {
          const uw_f4 u = cur_units[j];
          uw_f4 r;
          r.x = static_cast<float>(static_cast<double>(u.x) + shift[j % 3]);
          r.y = static_cast<float>(static_cast<double>(u.y) + shift[(j + 1) % 3]);
          r.z = static_cast<float>(static_cast<double>(u.z) + shift[(j + 2) % 3]);
          r.w = static_cast<float>(static_cast<double>(u.w) + shift[j % 3]);
          if (lane + 64 * j < units_here)
            __builtin_nontemporal_store(r, reinterpret_cast<uw_f4 *>(y) + u0 + lane + 64 * j);
        }
        if (last_local < kUwStep) {
          // the row's last frame lies in this step: words 3 last_local .. + 2
#pragma unroll
          for (int j = 0; j < kUnits; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const int g = (lane + 64 * j) * VEC + i - 3 * last_local;
              if (g >= 0 && g < 3) {
                last_pos[atom * 3 + g] = static_cast<double>(cur_units[j][i]);
                last_image[atom * 3 + g] = base[g];
              }
            }
        }
        tail[0] = low[0]; tail[1] = low[1]; tail[2] = low[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) before[d] = static_cast<double>(low[d]);
        continue;
      }
      tail[0] = low[0]; tail[1] = low[1]; tail[2] = low[2];
    }
#pragma unroll
    for (int j = 0; j < kUnits; ++j) *reinterpret_cast<Unit *>(mine + slot[j]) = cur_units[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float cur[kUwFrames][3];
    {
      float w[kUwWords];
#pragma unroll
      for (int j = 0; j < kUnits; ++j) {
        const Unit t = *reinterpret_cast<const Unit *>(mine + lane * kStride + j * VEC);
        if constexpr (VEC == 4) {
          w[4 * j] = t.x; w[4 * j + 1] = t.y; w[4 * j + 2] = t.z; w[4 * j + 3] = t.w;
        } else {
          w[j] = t;
        }
      }
#pragma unroll
      for (int f = 0; f < kUwFrames; ++f)
#pragma unroll
        for (int d = 0; d < 3; ++d) cur[f][d] = w[3 * f + d];
    }
    const int frames_here = static_cast<int>(min<int64_t>(kUwStep, n_frames - t0));
    const int fb = lane * kUwFrames;  // first frame of this lane in the step
    float res[kUwWords];
    // one component at a time: its jumps, their scan over the lanes, its results
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      // the frame before this lane's first one: the neighbour's last, or the carry
      const float left = __shfl_up(cur[kUwFrames - 1][d], 1);
      double prev = lane == 0 ? before[d] : static_cast<double>(left);
      double cum[kUwFrames];
      double run = 0.0;
#pragma unroll
      for (int f = 0; f < kUwFrames; ++f) {
        const double c = static_cast<double>(cur[f][d]);
        if (fb + f < frames_here)
          run = run + rint_quotient_near_zero(c - prev, box.v[d], inv_box[d]);
        cum[f] = run;
        prev = c;
      }
      double incl = run;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(incl, off);
        if (lane >= off) incl = incl + up;
      }
      const double preceding = incl - run;  // jumps of the lanes before this one
#pragma unroll
      for (int f = 0; f < kUwFrames; ++f) {
        const double c = static_cast<double>(cur[f][d]);
        const double img = -(cum[f] + preceding) + base[d];
        res[3 * f + d] = static_cast<float>(c + img * box.v[d]);
        if (fb + f == last_local) {
          last_pos[atom * 3 + d] = c;
          last_image[atom * 3 + d] = img;
        }
      }
      base[d] = base[d] - __shfl(incl, 63);
      // the step's last frame precedes the next step's first
      before[d] = static_cast<double>(__shfl(cur[kUwFrames - 1][d], 63));
    }
    if constexpr (VEC == 4) set_shift();
    // every lane has read its words: the image can take the results
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < kUnits; ++j) {
      Unit t;
      if constexpr (VEC == 4) {
        t = uw_f4{res[4 * j], res[4 * j + 1], res[4 * j + 2], res[4 * j + 3]};
      } else {
        t = res[j];
      }
      *reinterpret_cast<Unit *>(mine + lane * kStride + j * VEC) = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < kUnits; ++j) {
      const Unit t = *reinterpret_cast<const Unit *>(mine + slot[j]);
      if (lane + 64 * j < units_here)
        __builtin_nontemporal_store(t, reinterpret_cast<Unit *>(y) + u0 + lane + 64 * j);
    }
    // the reads above come before the next step's writes to the image
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// --------------------------------------------------------- element-wise maps

// A thread owns kMapUnits units of kPerThread consecutive elements (16-byte accesses for
// float32), a block's units lying kThreads units apart: every access of a wave is
// contiguous, and a thread has all its loads in flight before it computes (the float64
// arithmetic of one unit would otherwise sit between a block's loads and its stores and
// set the block's lifetime, which bounds the bytes in flight).
constexpr int kMapUnits = 4;

template <int OP, typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void map_kernel(const TIn *__restrict__ x,
                                                       const TIn *__restrict__ aux,
                                                       int64_t n_elements, int64_t n_frames,
                                                       Box3 par, Box3 inv,
                                                       TOut *__restrict__ out) {
  struct alignas(sizeof(TIn) * kPerThread) Pack { TIn v[kPerThread]; };
  struct alignas(sizeof(TOut) * kPerThread) PackOut { TOut v[kPerThread]; };
  constexpr int64_t kUnitStride = static_cast<int64_t>(kThreads) * kPerThread;
  const int64_t e_first =
      (static_cast<int64_t>(blockIdx.x) * kMapUnits * kThreads + threadIdx.x) * kPerThread;
  if (e_first >= n_elements) return;
  TIn in[kMapUnits][kPerThread], ax[kMapUnits][kPerThread];
#pragma unroll
  for (int u = 0; u < kMapUnits; ++u) {
    const int64_t e0 = e_first + u * kUnitStride;
    if (e0 + kPerThread <= n_elements) {
      const Pack p = *reinterpret_cast<const Pack *>(x + e0);
#pragma unroll
      for (int j = 0; j < kPerThread; ++j) in[u][j] = p.v[j];
      if (OP == MDX_MAP_UNWRAP_INDICES) {
        const Pack q = *reinterpret_cast<const Pack *>(aux + e0);
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) ax[u][j] = q.v[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < kPerThread; ++j) {
        in[u][j] = (e0 + j < n_elements) ? x[e0 + j] : TIn(0);
        if (OP == MDX_MAP_UNWRAP_INDICES)
          ax[u][j] = (e0 + j < n_elements) ? aux[e0 + j] : TIn(0);
      }
    }
  }
  // component and frame of the thread's first element; a unit stride is 1 mod 3
  const int d_first = static_cast<int>(e_first % 3);
  const int64_t t_first = (OP == MDX_MAP_VELOCITY) ? (e_first / 3) % n_frames : 0;
#pragma unroll
  for (int u = 0; u < kMapUnits; ++u) {
    const int64_t e0 = e_first + u * kUnitStride;
    if (e0 >= n_elements) break;
    const bool full = e0 + kPerThread <= n_elements;
    int d = (d_first + u) % 3;
    int64_t t = 0;
    if (OP == MDX_MAP_VELOCITY) {
      t = t_first + (u * kUnitStride + d_first) / 3;
      if (t >= n_frames) t %= n_frames;
    }
    // forward differences reach three words past the unit: the next lane's first words
    // (the wave's last lane and the array's last units read them from memory)
    TIn beyond[3] = {TIn(0), TIn(0), TIn(0)};
    if (OP == MDX_MAP_VELOCITY && full) {
#pragma unroll
      for (int i = 0; i < 3; ++i) beyond[i] = __shfl_down(in[u][i], 1);
      if ((threadIdx.x & 63) == 63 || e0 + 2 * kPerThread > n_elements) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          beyond[i] = (e0 + kPerThread + i < n_elements) ? x[e0 + kPerThread + i] : TIn(0);
      }
    }
    TOut res[kPerThread];
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      const double v = static_cast<double>(in[u][j]);
      double r;
      if (OP == MDX_MAP_WRAP) {
        r = v - floor_quotient(v, par.v[d], inv.v[d]) * par.v[d];
      } else if (OP == MDX_MAP_WRAP_CENTERED) {
        const double half = par.v[d] / 2.0;
        const double w = v + half;
        r = (w - floor_quotient(w, par.v[d], inv.v[d]) * par.v[d]) - half;
      } else if (OP == MDX_MAP_SCALE) {
        r = v * par.v[d];
      } else if (OP == MDX_MAP_UNWRAP_INDICES) {
        r = v + static_cast<double>(ax[u][j]) * par.v[d];
      } else {  // MDX_MAP_VELOCITY: forward difference, the last frame repeats the one before
        const int64_t e = e0 + j;
        if (e < n_elements) {
          if (t + 1 < n_frames) {
            const double next =
                !full ? static_cast<double>(x[e + 3])
                      : static_cast<double>(j + 3 < kPerThread ? in[u][j + 3]
                                                               : beyond[(j + 3) % kPerThread]);
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
  // float32 to float32 with enough atoms to fill the chip with one wave each: the
  // streaming kernel (MDX_UNWRAP_STREAM_MIN_ATOMS moves the threshold, for tests)
  int64_t stream_min_atoms = 512;
  if (const char *env = getenv("MDX_UNWRAP_STREAM_MIN_ATOMS")) {
    const long long v = atoll(env);
    if (v >= 1) stream_min_atoms = v;
  }
  const bool stream_kernel =
      dtype == MDX_F32 && out_dtype == MDX_F32 && n_atoms >= stream_min_atoms;
  {
  KernelTimer timer(MDX_SLOT_TRANSFORM, stream);
  if (stream_kernel) {
    const dim3 wave_grid(static_cast<unsigned>(ceil_div(n_atoms, static_cast<int64_t>(kUwWaves))));
    const bool rows_16b = n_frames % 4 == 0 && reinterpret_cast<uintptr_t>(dx) % 16 == 0 &&
                          reinterpret_cast<uintptr_t>(dout) % 16 == 0;
    if (rows_16b)
      hipLaunchKernelGGL(unwrap_stream_kernel<4>, wave_grid, dim3(kThreads), 0, stream,
                         static_cast<const float *>(dx), n_atoms, n_frames, b,
                         have_carry ? 1 : 0, d_last_pos.as<double>(),
                         d_last_image.as<double>(), static_cast<float *>(dout));
    else
      hipLaunchKernelGGL(unwrap_stream_kernel<1>, wave_grid, dim3(kThreads), 0, stream,
                         static_cast<const float *>(dx), n_atoms, n_frames, b,
                         have_carry ? 1 : 0, d_last_pos.as<double>(),
                         d_last_image.as<double>(), static_cast<float *>(dout));
  }
#define MDX_LAUNCH_UNWRAP(TIN, TOUT)                                                       \
  hipLaunchKernelGGL((unwrap_kernel<TIN, TOUT>), grid, dim3(kThreads), 0, stream,          \
                     static_cast<const TIN *>(dx), n_frames, b, have_carry ? 1 : 0,        \
                     d_last_pos.as<double>(), d_last_image.as<double>(),                   \
                     static_cast<TOUT *>(dout))
  if (stream_kernel) {
    // launched above
  } else if (dtype == MDX_F32) {
    if (out_dtype == MDX_F32) MDX_LAUNCH_UNWRAP(float, float); else MDX_LAUNCH_UNWRAP(float, double);
  } else {
    if (out_dtype == MDX_F32) MDX_LAUNCH_UNWRAP(double, float); else MDX_LAUNCH_UNWRAP(double, double);
  }
#undef MDX_LAUNCH_UNWRAP
  }
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
  const int64_t n_blocks = ceil_div(static_cast<int64_t>(n), kThreads * kPerThread * kMapUnits);
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
  {
  KernelTimer timer(MDX_SLOT_TRANSFORM, stream);
  switch (op) {
    case MDX_MAP_WRAP: MDX_LAUNCH_MAP(MDX_MAP_WRAP); break;
    case MDX_MAP_WRAP_CENTERED: MDX_LAUNCH_MAP(MDX_MAP_WRAP_CENTERED); break;
    case MDX_MAP_SCALE: MDX_LAUNCH_MAP(MDX_MAP_SCALE); break;
    case MDX_MAP_UNWRAP_INDICES: MDX_LAUNCH_MAP(MDX_MAP_UNWRAP_INDICES); break;
    default: MDX_LAUNCH_MAP(MDX_MAP_VELOCITY); break;
  }
  }
#undef MDX_LAUNCH_MAP
#undef MDX_LAUNCH_MAP_T
  MDX_HIP(hipGetLastError());
  if (location == MDX_DEVICE && out_location == MDX_DEVICE) return MDX_OK;  // stays queued
  if (out_location == MDX_HOST)
    MDX_HIP(hipMemcpyAsync(out, dout, n * dtype_size(out_dtype), hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

}  // extern "C"
