// Per-frame reductions over atoms: the streaming stage that turns per-atom
// trajectories [atoms][frames][k] into system series [frames][3] (ionic current,
// translational dipole moment, momentum flux, thermal flux).
//
// These kernels are bound by HBM bandwidth: every input element is read once and
// takes part in one multiply and one add.  A thread owns a few adjacent columns
// (frame, component) and walks down the atom axis, so a wave reads 1 KiB of
// contiguous memory per atom with eight independent loads in flight per lane.
// The atom axis is cut into chunks that run as separate blocks; their float64
// partial sums are combined by a second kernel in chunk order, so results do not
// depend on scheduling.
//
// The flux reductions combine arrays of different row widths per (atom, frame): stress
// [6], velocity [3], energies [1].  A lane that owned one frame would read 4-byte pieces
// 24 and 12 bytes apart; flux_stream_kernel instead has every wave fetch its 128 frames
// of each array with lane-contiguous 16- and 8-byte loads, passes the stress and the
// vector through a wave-private LDS image, and only then lets lane l pick up the two
// frames 2l, 2l+1 it accumulates.
//
// Compiled with -ffp-contract=off: products are rounded before they are added,
// as in the reference's separate multiply and reduce_sum operations.
#include "mdx_common.h"

using namespace mdx;

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;

template <typename T, int VEC>
struct VecLoad;
// every element is read exactly once: loads bypass the caches' retention (non-temporal)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
template <>
struct VecLoad<float, 4> {
  static __device__ __forceinline__ void load(const float *p, double (&v)[4]) {
    const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
template <>
struct VecLoad<double, 2> {
  static __device__ __forceinline__ void load(const double *p, double (&v)[2]) {
    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    v[0] = t.x; v[1] = t.y;
  }
};
template <typename T>
struct VecLoad<T, 1> {
  static __device__ __forceinline__ void load(const T *p, double (&v)[1]) {
    v[0] = static_cast<double>(*p);
  }
};

// partial[chunk][col] = sum over the chunk's atoms of w_a * x[a][col]
// kPick: only the components c >= pick_first of every group of pick_k columns are
// written, packed as [group][c - pick_first] (the momentum flux keeps components 3..5 of
// the six stress components; the whole 24-byte rows are fetched either way, the halves
// share their 64-byte sectors).
template <typename T, int VEC, bool kPick>
__global__ __launch_bounds__(kThreads) void weighted_sum_kernel(
    const T *__restrict__ x, const double *__restrict__ weights, double scalar,
    int64_t n_atoms, int64_t row_len, int64_t atoms_per_chunk, int pick_k, int pick_first,
    double *__restrict__ partial) {
  const int64_t col = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * VEC;
  if (col >= row_len) return;
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  double acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
  const T *p = x + a0 * row_len + col;
  int64_t a = a0;
  for (; a + kUnroll <= a1; a += kUnroll) {
    double vals[kUnroll][VEC];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) VecLoad<T, VEC>::load(p + u * row_len, vals[u]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const double w = weights ? weights[a + u] * scalar : scalar;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = acc[v] + w * vals[u][v];
    }
    p += kUnroll * row_len;
  }
  for (; a < a1; ++a) {
    double vals[VEC];
    VecLoad<T, VEC>::load(p, vals);
    const double w = weights ? weights[a] * scalar : scalar;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = acc[v] + w * vals[v];
    p += row_len;
  }
  if (!kPick) {
    double *dst = partial + static_cast<int64_t>(blockIdx.y) * row_len + col;
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[v] = acc[v];
  } else {
    const int kept = pick_k - pick_first;
    double *dst = partial + static_cast<int64_t>(blockIdx.y) * (row_len / pick_k) * kept;
    const int64_t group = col / pick_k;
    int c = static_cast<int>(col - group * pick_k);
    int64_t base = group * kept;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (c >= pick_first) dst[base + c - pick_first] = acc[v];
      if (++c == pick_k) {
        c = 0;
        base += kept;
      }
    }
  }
}

// Streaming flux reductions of float32 arrays with an even number of frames.
//   kThermal: partial[chunk][0][t*3+c] = sum_a (ke+pe) * v_c,
//             partial[chunk][1][t*3+c] = sum_a phi_c, phi = S . v for the symmetric stress
//             stored as xx,yy,zz,xy,xz,yz (six = stress, three = velocities);
//   else    : partial[chunk][t*3+c] = sum_a x_c * (ke+pe)   (three = x, six unused).
// Wave = 128 consecutive frames, walking down the chunk's atoms.  Per atom the wave's
// slice of every array is one contiguous range: 3072 B of stress (three 16-byte loads
// per lane), 1536 B of the vector (three 8-byte loads), 512 B of each energy (one 8-byte
// load, already the lane's own two frames).  Stress and vector are copied as they come
// into the wave's LDS image; lane l then reads words 12l..12l+11 and 6l..6l+5 (8-byte
// reads 12 and 6 words apart: sixteen lanes hit disjoint banks).  Only lanes of one wave
// exchange data, and a wave's LDS operations complete in order, so no barrier is needed.
// The next atom's loads are in flight while the current one is summed.  Every element
// is read once: non-temporal loads.
constexpr int kWaveFrames = 128;
typedef float f2v __attribute__((ext_vector_type(2)));

template <bool kThermal>
__global__ __launch_bounds__(kThreads) void flux_stream_kernel(
    const float *__restrict__ six, const float *__restrict__ three,
    const float *__restrict__ ke, const float *__restrict__ pe, int64_t n_atoms,
    int64_t n_frames, int64_t atoms_per_chunk, double *__restrict__ partial) {
  constexpr int kSixWords = kThermal ? kWaveFrames * 6 : 0;
  constexpr int kWaves = kThreads / 64;
  __shared__ __align__(16) float image[kWaves][kSixWords + kWaveFrames * 3];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t t0 = (static_cast<int64_t>(blockIdx.x) * kWaves + wave) * kWaveFrames;
  if (t0 >= n_frames) return;  // no block-wide synchronisation below
  const int frames_here = static_cast<int>(min<int64_t>(kWaveFrames, n_frames - t0));  // even
  const int six_units = frames_here * 6 / 4;   // 16-byte units of the stress slice
  const int three_units = frames_here * 3 / 2; // 8-byte units of the vector slice
  const int pair_units = frames_here / 2;      // 8-byte units of an energy slice
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  float *my_six = image[wave];
  float *my_three = image[wave] + kSixWords;

  f4v s_next[3];
  f2v v_next[3], k_next, p_next;
  // Units past the slice's end (last tile of the frame axis) re-read the slice's last unit
  // and are replaced by zero where the values are used (not here: a select right behind
  // the load would make the wave wait for it): no branch around a load.
  auto fetch = [&](int64_t a) {
    const int64_t at = a * n_frames + t0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int u = lane + 64 * j;
      if (kThermal)
        s_next[j] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(six + at * 6) +
                                               min(u, six_units - 1));
      v_next[j] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(three + at * 3) +
                                             min(u, three_units - 1));
    }
    const int own = min(lane, pair_units - 1);
    k_next = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(ke + at) + own);
    p_next = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(pe + at) + own);
  };

  double ev[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  double ph[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (a0 < a1) fetch(a0);
  for (int64_t a = a0; a < a1; ++a) {
    f4v s_cur[3];
    f2v v_cur[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int u = lane + 64 * j;
      if (kThermal) s_cur[j] = u < six_units ? s_next[j] : f4v{0.f, 0.f, 0.f, 0.f};
      v_cur[j] = u < three_units ? v_next[j] : f2v{0.f, 0.f};
    }
    const f2v k_cur = lane < pair_units ? k_next : f2v{0.f, 0.f};
    const f2v p_cur = lane < pair_units ? p_next : f2v{0.f, 0.f};
    if (a + 1 < a1) fetch(a + 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (kThermal) *reinterpret_cast<f4v *>(my_six + 4 * (lane + 64 * j)) = s_cur[j];
      *reinterpret_cast<f2v *>(my_three + 2 * (lane + 64 * j)) = v_cur[j];
    }
    // the wave's own writes are visible to its later reads; keep the compiler from
    // moving the reads of other lanes' words above them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float sw[12], vw[6];
    if (kThermal) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f2v t = *reinterpret_cast<const f2v *>(my_six + 12 * lane + 2 * j);
        sw[2 * j] = t.x;
        sw[2 * j + 1] = t.y;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f2v t = *reinterpret_cast<const f2v *>(my_three + 6 * lane + 2 * j);
      vw[2 * j] = t.x;
      vw[2 * j + 1] = t.y;
    }
    // reads done before the next atom's image overwrites them (program order within the
    // wave; the fence keeps the compiler from sinking the reads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      double v[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) v[i] = static_cast<double>(vw[3 * f + i]);
      const double e = static_cast<double>(f == 0 ? k_cur.x : k_cur.y) +
                       static_cast<double>(f == 0 ? p_cur.x : p_cur.y);
      if (kThermal) {
        double s[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = static_cast<double>(sw[6 * f + i]);
        ph[f][0] = ph[f][0] + ((s[0] * v[0] + s[3] * v[1]) + s[4] * v[2]);
        ph[f][1] = ph[f][1] + ((s[3] * v[0] + s[1] * v[1]) + s[5] * v[2]);
        ph[f][2] = ph[f][2] + ((s[4] * v[0] + s[5] * v[1]) + s[2] * v[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) ev[f][i] = ev[f][i] + e * v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) ev[f][i] = ev[f][i] + v[i] * e;
      }
    }
  }
  if (lane >= pair_units) return;
  const int64_t n_cols = n_frames * 3;
  double *dst = partial + static_cast<int64_t>(blockIdx.y) * (kThermal ? 2 : 1) * n_cols +
                (t0 + 2 * lane) * 3;
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dst[3 * f + i] = ev[f][i];
      if (kThermal) dst[n_cols + 3 * f + i] = ph[f][i];
    }
}

// partial[chunk][0][t*3+c] = sum_a (ke+pe) * v_c ; partial[chunk][1][t*3+c] = sum_a phi_c
// with phi = S . v for the symmetric stress stored as xx,yy,zz,xy,xz,yz.
template <typename T>
__global__ __launch_bounds__(kThreads) void thermal_flux_kernel(
    const T *__restrict__ stress, const T *__restrict__ vel, const T *__restrict__ ke,
    const T *__restrict__ pe, int64_t n_atoms, int64_t n_frames, int64_t atoms_per_chunk,
    double *__restrict__ partial) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= n_frames) return;
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  double ev[3] = {0.0, 0.0, 0.0};
  double ph[3] = {0.0, 0.0, 0.0};
  for (int64_t a = a0; a < a1; ++a) {
    const int64_t at = a * n_frames + t;
    double s[6], v[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = static_cast<double>(stress[at * 6 + i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = static_cast<double>(vel[at * 3 + i]);
    const double e = static_cast<double>(ke[at]) + static_cast<double>(pe[at]);
    ph[0] = ph[0] + ((s[0] * v[0] + s[3] * v[1]) + s[4] * v[2]);
    ph[1] = ph[1] + ((s[3] * v[0] + s[1] * v[1]) + s[5] * v[2]);
    ph[2] = ph[2] + ((s[4] * v[0] + s[5] * v[1]) + s[2] * v[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) ev[i] = ev[i] + e * v[i];
  }
  const int64_t n_cols = n_frames * 3;
  double *dst = partial + static_cast<int64_t>(blockIdx.y) * 2 * n_cols;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dst[t * 3 + i] = ev[i];
    dst[n_cols + t * 3 + i] = ph[i];
  }
}

// partial[chunk][t*3+c] = sum_a x[a][t][c] * (ke[a][t] + pe[a][t]); the energies are
// added first, then multiplied, as in the reference expression.
template <typename T>
__global__ __launch_bounds__(kThreads) void energy_weighted_kernel(
    const T *__restrict__ x, const T *__restrict__ ke, const T *__restrict__ pe,
    int64_t n_atoms, int64_t n_frames, int64_t atoms_per_chunk,
    double *__restrict__ partial) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= n_frames) return;
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  double acc[3] = {0.0, 0.0, 0.0};
  for (int64_t a = a0; a < a1; ++a) {
    const int64_t at = a * n_frames + t;
    const double e = static_cast<double>(ke[at]) + static_cast<double>(pe[at]);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = acc[i] + static_cast<double>(x[at * 3 + i]) * e;
  }
  double *dst = partial + static_cast<int64_t>(blockIdx.y) * n_frames * 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) dst[t * 3 + i] = acc[i];
}

// ---- Kinaci integrated heat current: a running time integral per atom in front of
// the atom reduction.  Thread = frame, block = 256 consecutive frames, grid.y = atom
// chunks.  The power force.velocity of four atoms at a time is scanned over the block's
// frames (wave shuffles, then the four waves' totals through LDS); a first launch
// leaves every block's total per atom, a per-atom pass turns those into the integral
// at each block's first frame, and a second launch of the same scan adds that offset
// and reduces position * (integral + potential energy) over the chunk's atoms.
constexpr int kScanAtoms = 4;

__device__ __forceinline__ void block_inclusive_scan(double (&v)[kScanAtoms],
                                                     double (*wave_tot)[kScanAtoms]) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
#pragma unroll
    for (int u = 0; u < kScanAtoms; ++u) {
      const double below = __shfl_up(v[u], off);
      if (lane >= off) v[u] = v[u] + below;
    }
  __syncthreads();  // the totals of the previous four atoms have been read
  if (lane == 63)
#pragma unroll
    for (int u = 0; u < kScanAtoms; ++u) wave_tot[wave][u] = v[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kScanAtoms; ++u) {
    double before = 0.0;
    for (int w = 0; w < wave; ++w) before = before + wave_tot[w][u];
    v[u] = v[u] + before;
  }
}

// kReduce == false: totals[block][atom] = sum over the block's frames of force.vel
// kReduce == true:  partial[chunk][t*3+c] = sum_a pos[a][t][c] * (carry[a] + dt *
//                   (totals[block][a] + scan) + pe[a][t]), totals holding the sums of
//                   the blocks before this one
template <typename T, bool kReduce>
__global__ __launch_bounds__(kThreads) void kinaci_kernel(
    const T *__restrict__ pos, const T *__restrict__ vel, const T *__restrict__ force,
    const T *__restrict__ pe, int64_t n_atoms, int64_t n_frames, int64_t atoms_per_chunk,
    double dt, const double *__restrict__ carry, double *__restrict__ totals,
    double *__restrict__ partial) {
  __shared__ double wave_tot[kThreads / 64][kScanAtoms];
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const bool live = t < n_frames;
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  double acc[3] = {0.0, 0.0, 0.0};
  for (int64_t a = a0; a < a1; a += kScanAtoms) {
    double power[kScanAtoms];
#pragma unroll
    for (int u = 0; u < kScanAtoms; ++u) {
      power[u] = 0.0;
      if (live && a + u < a1) {
        const int64_t at = ((a + u) * n_frames + t) * 3;
        double sum = static_cast<double>(force[at]) * static_cast<double>(vel[at]);
        sum = sum + static_cast<double>(force[at + 1]) * static_cast<double>(vel[at + 1]);
        sum = sum + static_cast<double>(force[at + 2]) * static_cast<double>(vel[at + 2]);
        power[u] = sum;
      }
    }
    block_inclusive_scan(power, wave_tot);
    if (!kReduce) {
      if (threadIdx.x == kThreads - 1)
#pragma unroll
        for (int u = 0; u < kScanAtoms; ++u)
          if (a + u < a1) totals[static_cast<int64_t>(blockIdx.x) * n_atoms + a + u] = power[u];
      continue;
    }
    if (!live) continue;
#pragma unroll
    for (int u = 0; u < kScanAtoms; ++u) {
      if (a + u >= a1) continue;
      const int64_t at = (a + u) * n_frames + t;
      const double before = totals[static_cast<int64_t>(blockIdx.x) * n_atoms + a + u];
      const double integral = carry[a + u] + (before + power[u]) * dt;
      const double e = integral + static_cast<double>(pe[at]);
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = acc[i] + static_cast<double>(pos[at * 3 + i]) * e;
    }
  }
  if (kReduce && live) {
    double *dst = partial + static_cast<int64_t>(blockIdx.y) * n_frames * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[t * 3 + i] = acc[i];
  }
}

// The same for float32 arrays with an even number of frames, as a stream in the manner of
// flux_stream_kernel: wave = 128 consecutive frames (a slice), walking down the chunk's
// atoms; position, velocity and force pass through the wave's LDS image, lane l then owns
// the frames 2l, 2l+1.  The integral is split at the slice's start,
//   I_a(t) = carry_a + dt * (sum of the power before the slice) + dt * (sum within it),
// so that one pass over all four arrays needs nothing from other slices: it adds
// pos * (carry + dt * within + pe) and leaves the slice's power sum per atom in `totals`;
// the part pos * dt * before follows as a column sum of the positions alone, weighted per
// slice and atom (sliced_weighted_sum_kernel), once kinaci_offsets_kernel has turned the
// sums into the sums before each slice.  13 words per atom and frame are fetched for the
// 10 that are needed (the two-launch scan above fetches 16).
__global__ __launch_bounds__(kThreads) void kinaci_stream_kernel(
    const float *__restrict__ pos, const float *__restrict__ vel,
    const float *__restrict__ force, const float *__restrict__ pe, int64_t n_atoms,
    int64_t n_frames, int64_t atoms_per_chunk, double dt, const double *__restrict__ carry,
    double *__restrict__ totals, double *__restrict__ partial) {
  constexpr int kWaves = kThreads / 64;
  constexpr int kWords = kWaveFrames * 3;
  __shared__ __align__(16) float image[kWaves][3 * kWords];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t slice = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  const int64_t t0 = slice * kWaveFrames;
  if (t0 >= n_frames) return;  // no block-wide synchronisation below
  const int frames_here = static_cast<int>(min<int64_t>(kWaveFrames, n_frames - t0));  // even
  const int three_units = frames_here * 3 / 2;  // 8-byte units of a vector slice
  const int pair_units = frames_here / 2;       // 8-byte units of the energy slice
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  float *my_pos = image[wave];
  float *my_vel = image[wave] + kWords;
  float *my_force = image[wave] + 2 * kWords;

  f2v x_next[3], v_next[3], f_next[3], p_next;
  // units past the slice's end re-read its last unit and become zero where they are used
  auto fetch = [&](int64_t a) {
    const int64_t at = a * n_frames + t0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int u = min(lane + 64 * j, three_units - 1);
      x_next[j] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(pos + at * 3) + u);
      v_next[j] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(vel + at * 3) + u);
      f_next[j] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(force + at * 3) + u);
    }
    p_next = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(pe + at) +
                                        min(lane, pair_units - 1));
  };

  double acc[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (a0 < a1) fetch(a0);
  for (int64_t a = a0; a < a1; ++a) {
    const f2v zero2 = f2v{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int u = lane + 64 * j;
      const bool live = u < three_units;
      *reinterpret_cast<f2v *>(my_pos + 2 * u) = live ? x_next[j] : zero2;
      *reinterpret_cast<f2v *>(my_vel + 2 * u) = live ? v_next[j] : zero2;
      *reinterpret_cast<f2v *>(my_force + 2 * u) = live ? f_next[j] : zero2;
    }
    const f2v p_cur = lane < pair_units ? p_next : zero2;
    const double carry_a = carry[a];
    if (a + 1 < a1) fetch(a + 1);
    // the wave's own writes are visible to its later reads; keep the compiler from moving
    // the reads of other lanes' words above them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float xw[6], vw[6], fw[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f2v tx = *reinterpret_cast<const f2v *>(my_pos + 6 * lane + 2 * j);
      const f2v tv = *reinterpret_cast<const f2v *>(my_vel + 6 * lane + 2 * j);
      const f2v tf = *reinterpret_cast<const f2v *>(my_force + 6 * lane + 2 * j);
      xw[2 * j] = tx.x; xw[2 * j + 1] = tx.y;
      vw[2 * j] = tv.x; vw[2 * j + 1] = tv.y;
      fw[2 * j] = tf.x; fw[2 * j + 1] = tf.y;
    }
    // reads done before the next atom's image overwrites them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double power[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      double sum = static_cast<double>(fw[3 * f]) * static_cast<double>(vw[3 * f]);
      sum = sum + static_cast<double>(fw[3 * f + 1]) * static_cast<double>(vw[3 * f + 1]);
      sum = sum + static_cast<double>(fw[3 * f + 2]) * static_cast<double>(vw[3 * f + 2]);
      power[f] = sum;  // zero for frames past the end: their words are zero
    }
    const double own = power[0] + power[1];
    double incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double below = __shfl_up(incl, off);
      if (lane >= off) incl = incl + below;
    }
    if (lane == 63) totals[slice * n_atoms + a] = incl;
    const double before_lane = incl - own;
    const double within[2] = {before_lane + power[0], before_lane + own};
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const double integral = carry_a + within[f] * dt;
      const double e = integral + static_cast<double>(f == 0 ? p_cur.x : p_cur.y);
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[f][i] = acc[f][i] + static_cast<double>(xw[3 * f + i]) * e;
    }
  }
  if (lane >= pair_units) return;
  double *dst = partial + static_cast<int64_t>(blockIdx.y) * n_frames * 3 + (t0 + 2 * lane) * 3;
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[3 * f + i] = acc[f][i];
}

// partial[chunk][col] = scalar * sum over the chunk's atoms of w[col / cols_per_slice][a] *
// x[a][col]: the column sum of weighted_sum_kernel with a weight per (slice of columns,
// atom).  cols_per_slice is a multiple of four, so a thread's columns share their weight.
__global__ __launch_bounds__(kThreads) void sliced_weighted_sum_kernel(
    const float *__restrict__ x, const double *__restrict__ weights, double scalar,
    int64_t n_atoms, int64_t row_len, int64_t cols_per_slice, int64_t atoms_per_chunk,
    double *__restrict__ partial) {
  const int64_t col = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * 4;
  if (col >= row_len) return;
  const int64_t a0 = static_cast<int64_t>(blockIdx.y) * atoms_per_chunk;
  const int64_t a1 = min(n_atoms, a0 + atoms_per_chunk);
  const double *w = weights + (col / cols_per_slice) * n_atoms;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const float *p = x + a0 * row_len + col;
  int64_t a = a0;
  for (; a + kUnroll <= a1; a += kUnroll) {
    double vals[kUnroll][4];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) VecLoad<float, 4>::load(p + u * row_len, vals[u]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const double wa = w[a + u] * scalar;
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] = acc[v] + wa * vals[u][v];
    }
    p += kUnroll * row_len;
  }
  for (; a < a1; ++a) {
    double vals[4];
    VecLoad<float, 4>::load(p, vals);
    const double wa = w[a] * scalar;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = acc[v] + wa * vals[v];
    p += row_len;
  }
  double *dst = partial + static_cast<int64_t>(blockIdx.y) * row_len + col;
#pragma unroll
  for (int v = 0; v < 4; ++v) dst[v] = acc[v];
}

// Per atom: totals[b][a] becomes the sum of the blocks before b; carry[a] moves on to
// the integral at the batch's last frame (carry_next), the old value staying in `carry`
// for the second launch.
// Blocks of one wave: a thread per atom leaves few waves, so they are spread over the chip.
__global__ __launch_bounds__(64) void kinaci_offsets_kernel(
    double *__restrict__ totals, int64_t n_blocks, int64_t n_atoms, double dt,
    const double *__restrict__ carry, double *__restrict__ carry_next) {
  const int64_t a = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= n_atoms) return;
  double run = 0.0;
  int64_t b = 0;
  // eight loads in flight: the entries of an atom lie n_atoms apart and each is read
  // before it is overwritten
  for (; b + 8 <= n_blocks; b += 8) {
    double own[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) own[u] = totals[(b + u) * n_atoms + a];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      totals[(b + u) * n_atoms + a] = run;
      run = run + own[u];
    }
  }
  for (; b < n_blocks; ++b) {
    const double own = totals[b * n_atoms + a];
    totals[b * n_atoms + a] = run;
    run = run + own;
  }
  carry_next[a] = carry[a] + run * dt;
}

// out[col] (+)= sum over chunks, in chunk order.  planes == 2 combines the two
// partial sums of the thermal flux as (sum of plane 0) - (sum of plane 1).
__global__ __launch_bounds__(kThreads) void combine_chunks_kernel(
    const double *__restrict__ partial, int n_chunks, int planes, int64_t n_cols,
    double *__restrict__ out, int accumulate) {
  const int64_t col = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (col >= n_cols) return;
  double s0 = 0.0, s1 = 0.0;
  for (int c = 0; c < n_chunks; ++c) {
    const double *p = partial + static_cast<int64_t>(c) * planes * n_cols;
    s0 = s0 + p[col];
    if (planes == 2) s1 = s1 + p[n_cols + col];
  }
  const double r = (planes == 2) ? (s0 - s1) : s0;
  out[col] = accumulate ? (out[col] + r) : r;
}

// Partial sums of mdx_weighted_atom_sum_static: one copy of this array per GPU for the
// life of the process (32 MiB).  In host code the name is only a handle for
// hipGetSymbolAddress; kernels receive the address obtained there.
constexpr int64_t kScratchDoubles = int64_t(4) << 20;
static __device__ __align__(256) double scratch[kScratchDoubles];

constexpr int kScratchDevices = 64;
std::mutex g_scratch_mutex;
double *g_scratch_ptr[kScratchDevices] = {nullptr};

// Address of the current device's copy of `scratch` (looked up once per device).
int scratch_address(double **ptr) {
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kScratchDevices)
    return fail(MDX_ERR_DEVICE, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void *address = nullptr;
    MDX_HIP(hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)));
    if (address == nullptr) return fail(MDX_ERR_DEVICE, "hipGetSymbolAddress returned null");
    g_scratch_ptr[dev] = static_cast<double *>(address);
  }
  *ptr = g_scratch_ptr[dev];
  return MDX_OK;
}

struct ChunkPlan {
  int64_t col_blocks;
  int64_t n_chunks;
  int64_t atoms_per_chunk;
};

ChunkPlan plan_chunks(int64_t n_atoms, int64_t threads_needed) {
  ChunkPlan p;
  p.col_blocks = ceil_div(threads_needed, kThreads);
  int64_t want = ceil_div(4096, p.col_blocks);          // aim for >= 4096 blocks
  int64_t max_chunks = ceil_div(n_atoms, 4 * kUnroll);  // at least 32 atoms per chunk
  if (max_chunks < 1) max_chunks = 1;
  p.n_chunks = want < max_chunks ? want : max_chunks;
  if (p.n_chunks < 1) p.n_chunks = 1;
  if (p.n_chunks > 65535) p.n_chunks = 65535;
  p.atoms_per_chunk = ceil_div(n_atoms, p.n_chunks);
  if (p.atoms_per_chunk < 1) p.atoms_per_chunk = 1;
  p.n_chunks = ceil_div(n_atoms, p.atoms_per_chunk);
  if (p.n_chunks < 1) p.n_chunks = 1;
  return p;
}

// Writes the combined result to `out` (device or host).
int finish(const DeviceBuffer &partial, int n_chunks, int planes, int64_t n_cols,
           double *out, int out_location, int accumulate, hipStream_t stream) {
  const unsigned blocks = static_cast<unsigned>(ceil_div(n_cols, kThreads));
  if (out_location == MDX_DEVICE) {
    hipLaunchKernelGGL(combine_chunks_kernel, dim3(blocks), dim3(kThreads), 0, stream,
                       partial.as<double>(), n_chunks, planes, n_cols, out, accumulate);
    MDX_HIP(hipGetLastError());
    return MDX_OK;
  }
  DeviceBuffer result;
  MDX_TRY(result.alloc(sizeof(double) * static_cast<size_t>(n_cols)));
  hipLaunchKernelGGL(combine_chunks_kernel, dim3(blocks), dim3(kThreads), 0, stream,
                     partial.as<double>(), n_chunks, planes, n_cols, result.as<double>(), 0);
  MDX_HIP(hipGetLastError());
  std::vector<double> host(static_cast<size_t>(n_cols));
  MDX_HIP(hipMemcpyAsync(host.data(), result.ptr, sizeof(double) * host.size(),
                         hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  for (int64_t i = 0; i < n_cols; ++i) out[i] = accumulate ? out[i] + host[i] : host[i];
  return MDX_OK;
}

int check_common(const void *x, int dtype, int location, int64_t n_atoms, int64_t n_frames,
                 const double *out, int out_location) {
  MDX_REQUIRE(x != nullptr && out != nullptr, "null argument");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  MDX_REQUIRE(out_location == MDX_HOST || out_location == MDX_DEVICE, "bad out_location");
  MDX_REQUIRE(n_atoms >= 0 && n_frames >= 0, "negative size");
  return MDX_OK;
}

int zero_result(double *out, int out_location, int64_t n_cols, int accumulate,
                hipStream_t stream) {
  if (accumulate || n_cols == 0) return MDX_OK;
  if (out_location == MDX_HOST) {
    memset(out, 0, sizeof(double) * static_cast<size_t>(n_cols));
  } else {
    MDX_HIP(hipMemsetAsync(out, 0, sizeof(double) * static_cast<size_t>(n_cols), stream));
  }
  return MDX_OK;
}

// Launches the weighted column sum (all columns, or the picked components of every group
// of pick_k columns when pick_k > 0) of x [n_atoms][row_len] into partial[chunk][...].
// `fit_chunks` > 0 caps the number of atom chunks (a caller with a fixed scratch array).
int launch_column_sum(const void *x, int dtype, const double *weights, double scalar,
                      int64_t n_atoms, int64_t row_len, int pick_k, int pick_first,
                      int64_t fit_chunks, double *partial_fixed, DeviceBuffer *partial,
                      ChunkPlan *plan_out, hipStream_t stream) {
  const int width = (dtype == MDX_F32) ? 4 : 2;
  const bool wide = (row_len % width == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  const int64_t threads_needed = wide ? row_len / width : row_len;
  ChunkPlan plan = plan_chunks(n_atoms, threads_needed);
  if (fit_chunks > 0 && plan.n_chunks > fit_chunks) {
    plan.atoms_per_chunk = ceil_div(n_atoms, fit_chunks);
    plan.n_chunks = ceil_div(n_atoms, plan.atoms_per_chunk);
  }
  const int64_t out_cols = pick_k > 0 ? (row_len / pick_k) * (pick_k - pick_first) : row_len;
  double *dst = partial_fixed;
  if (dst == nullptr) {
    MDX_TRY(partial->alloc(sizeof(double) * static_cast<size_t>(plan.n_chunks) * out_cols));
    dst = partial->as<double>();
  }
  dim3 grid(static_cast<unsigned>(plan.col_blocks), static_cast<unsigned>(plan.n_chunks));
#define MDX_COLUMN_SUM(TYPE, VEC, PICK)                                                     \
  hipLaunchKernelGGL((weighted_sum_kernel<TYPE, VEC, PICK>), grid, dim3(kThreads), 0,       \
                     stream, static_cast<const TYPE *>(x), weights, scalar, n_atoms,        \
                     row_len, plan.atoms_per_chunk, pick_k, pick_first, dst)
  if (dtype == MDX_F32) {
    if (wide) {
      if (pick_k > 0) MDX_COLUMN_SUM(float, 4, true); else MDX_COLUMN_SUM(float, 4, false);
    } else {
      if (pick_k > 0) MDX_COLUMN_SUM(float, 1, true); else MDX_COLUMN_SUM(float, 1, false);
    }
  } else {
    if (wide) {
      if (pick_k > 0) MDX_COLUMN_SUM(double, 2, true); else MDX_COLUMN_SUM(double, 2, false);
    } else {
      if (pick_k > 0) MDX_COLUMN_SUM(double, 1, true); else MDX_COLUMN_SUM(double, 1, false);
    }
  }
#undef MDX_COLUMN_SUM
  MDX_HIP(hipGetLastError());
  *plan_out = plan;
  return MDX_OK;
}

// True when the streaming flux kernel applies: float32, an even number of frames (every
// row then starts on the load width of its array) and 16-byte aligned arrays.
bool stream_layout_ok(int dtype, int64_t n_frames, std::initializer_list<const void *> arrays) {
  if (dtype != MDX_F32 || (n_frames & 1) != 0) return false;
  for (const void *p : arrays)
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) return false;
  return true;
}

ChunkPlan plan_stream_chunks(int64_t n_atoms, int64_t n_frames) {
  // one wave per 128 frames, four waves per block
  return plan_chunks(n_atoms, ceil_div(n_frames, kWaveFrames) * 64);
}

}  // namespace

extern "C" {

int mdx_weighted_atom_sum(const void *x, int dtype, int x_location, int64_t n_atoms,
                          int64_t n_frames, int k, const double *weights,
                          double scalar_weight, double *out, int out_location,
                          int accumulate) {
  MDX_TRY(check_common(x, dtype, x_location, n_atoms, n_frames, out, out_location));
  MDX_REQUIRE(k >= 1 && k <= 64, "k out of range");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const int64_t row_len = n_frames * k;
  if (n_atoms == 0 || row_len == 0) return zero_result(out, out_location, row_len, accumulate, stream);

  DeviceBuffer x_staged, w_dev, partial;
  const void *dx = nullptr;
  MDX_TRY(to_device(x, x_location, static_cast<size_t>(n_atoms) * row_len * dtype_size(dtype),
                    x_staged, stream, &dx));
  const double *dw = nullptr;
  if (weights != nullptr) {
    MDX_TRY(w_dev.alloc(sizeof(double) * static_cast<size_t>(n_atoms)));
    MDX_HIP(hipMemcpyAsync(w_dev.ptr, weights, sizeof(double) * n_atoms,
                           hipMemcpyHostToDevice, stream));
    dw = w_dev.as<double>();
  }
  ChunkPlan plan;
  {
    KernelTimer timer(MDX_SLOT_ATOM_SUM, stream);
    MDX_TRY(launch_column_sum(dx, dtype, dw, scalar_weight, n_atoms, row_len, 0, 0, 0, nullptr,
                              &partial, &plan, stream));
  }
  MDX_TRY(finish(partial, static_cast<int>(plan.n_chunks), 1, row_len, out, out_location,
                 accumulate, stream));
  // Device to device: the work stays queued.  The scoped device buffers go back to the
  // library's cache on return, and whoever takes them next queues on this same stream.
  if (x_location == MDX_DEVICE && out_location == MDX_DEVICE && weights == nullptr)
    return MDX_OK;
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

int mdx_weighted_atom_sum_static(const void *x, int dtype, int64_t n_atoms, int64_t n_frames,
                                 int k, double scalar_weight, double *out, int accumulate) {
  MDX_REQUIRE(x != nullptr && out != nullptr, "null argument");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(n_atoms >= 0 && n_frames >= 0, "negative size");
  MDX_REQUIRE(k >= 1 && k <= 64, "k out of range");
  const int64_t row_len = n_frames * k;
  MDX_REQUIRE(row_len <= kScratchDoubles,
              "n_frames * k = %lld exceeds the %lld values the static scratch array holds; "
              "split the frames over several calls or use mdx_weighted_atom_sum",
              (long long)row_len, (long long)kScratchDoubles);
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  double *partial = nullptr;
  MDX_TRY(scratch_address(&partial));
  if (n_atoms == 0 || row_len == 0) return zero_result(out, MDX_DEVICE, row_len, accumulate, stream);

  ChunkPlan plan;
  MDX_TRY(launch_column_sum(x, dtype, nullptr, scalar_weight, n_atoms, row_len, 0, 0,
                            kScratchDoubles / row_len, partial, nullptr, &plan, stream));
  hipLaunchKernelGGL(combine_chunks_kernel,
                     dim3(static_cast<unsigned>(ceil_div(row_len, kThreads))), dim3(kThreads), 0,
                     stream, partial, static_cast<int>(plan.n_chunks), 1, row_len, out,
                     accumulate);
  MDX_HIP(hipGetLastError());
  return MDX_OK;
}

int mdx_momentum_flux(const void *stress, int dtype, int location, int64_t n_atoms,
                      int64_t n_frames, double *out, int out_location, int accumulate) {
  MDX_TRY(check_common(stress, dtype, location, n_atoms, n_frames, out, out_location));
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const int64_t n_cols = n_frames * 3;
  if (n_atoms == 0 || n_cols == 0) return zero_result(out, out_location, n_cols, accumulate, stream);
  DeviceBuffer staged, partial;
  const void *ds = nullptr;
  MDX_TRY(to_device(stress, location,
                    static_cast<size_t>(n_atoms) * n_frames * 6 * dtype_size(dtype), staged,
                    stream, &ds));
  // the column sum of the [n_frames][6] rows, of which components 3..5 are kept
  ChunkPlan plan;
  {
    KernelTimer timer(MDX_SLOT_FLUX, stream);
    MDX_TRY(launch_column_sum(ds, dtype, nullptr, 1.0, n_atoms, n_frames * 6, 6, 3, 0, nullptr,
                              &partial, &plan, stream));
  }
  MDX_TRY(finish(partial, static_cast<int>(plan.n_chunks), 1, n_cols, out, out_location,
                 accumulate, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

int mdx_thermal_flux(const void *stress, const void *vel, const void *ke, const void *pe,
                     int dtype, int location, int64_t n_atoms, int64_t n_frames,
                     double *out, int out_location, int accumulate) {
  MDX_TRY(check_common(stress, dtype, location, n_atoms, n_frames, out, out_location));
  MDX_REQUIRE(vel != nullptr && ke != nullptr && pe != nullptr, "null argument");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const int64_t n_cols = n_frames * 3;
  if (n_atoms == 0 || n_cols == 0) return zero_result(out, out_location, n_cols, accumulate, stream);
  DeviceBuffer st_s, st_v, st_k, st_p, partial;
  const void *ds = nullptr, *dv = nullptr, *dk = nullptr, *dp = nullptr;
  const size_t unit = static_cast<size_t>(n_atoms) * n_frames * dtype_size(dtype);
  MDX_TRY(to_device(stress, location, unit * 6, st_s, stream, &ds));
  MDX_TRY(to_device(vel, location, unit * 3, st_v, stream, &dv));
  MDX_TRY(to_device(ke, location, unit, st_k, stream, &dk));
  MDX_TRY(to_device(pe, location, unit, st_p, stream, &dp));
  const bool streamed = stream_layout_ok(dtype, n_frames, {ds, dv, dk, dp});
  const ChunkPlan plan =
      streamed ? plan_stream_chunks(n_atoms, n_frames) : plan_chunks(n_atoms, n_frames);
  MDX_TRY(partial.alloc(sizeof(double) * static_cast<size_t>(plan.n_chunks) * 2 * n_cols));
  dim3 grid(static_cast<unsigned>(plan.col_blocks), static_cast<unsigned>(plan.n_chunks));
  {
  KernelTimer timer(MDX_SLOT_FLUX, stream);
  if (streamed)
    hipLaunchKernelGGL(flux_stream_kernel<true>, grid, dim3(kThreads), 0, stream,
                       static_cast<const float *>(ds), static_cast<const float *>(dv),
                       static_cast<const float *>(dk), static_cast<const float *>(dp),
                       n_atoms, n_frames, plan.atoms_per_chunk, partial.as<double>());
  else if (dtype == MDX_F32)
    hipLaunchKernelGGL(thermal_flux_kernel<float>, grid, dim3(kThreads), 0, stream,
                       static_cast<const float *>(ds), static_cast<const float *>(dv),
                       static_cast<const float *>(dk), static_cast<const float *>(dp),
                       n_atoms, n_frames, plan.atoms_per_chunk, partial.as<double>());
  else
    hipLaunchKernelGGL(thermal_flux_kernel<double>, grid, dim3(kThreads), 0, stream,
                       static_cast<const double *>(ds), static_cast<const double *>(dv),
                       static_cast<const double *>(dk), static_cast<const double *>(dp),
                       n_atoms, n_frames, plan.atoms_per_chunk, partial.as<double>());
  }
  MDX_HIP(hipGetLastError());
  MDX_TRY(finish(partial, static_cast<int>(plan.n_chunks), 2, n_cols, out, out_location,
                 accumulate, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

int mdx_energy_weighted_sum(const void *x, const void *ke, const void *pe, int dtype,
                            int location, int64_t n_atoms, int64_t n_frames, double *out,
                            int out_location, int accumulate) {
  MDX_TRY(check_common(x, dtype, location, n_atoms, n_frames, out, out_location));
  MDX_REQUIRE(ke != nullptr && pe != nullptr, "null argument");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const int64_t n_cols = n_frames * 3;
  if (n_atoms == 0 || n_cols == 0) return zero_result(out, out_location, n_cols, accumulate, stream);
  DeviceBuffer st_x, st_k, st_p, partial;
  const void *dx = nullptr, *dk = nullptr, *dp = nullptr;
  const size_t unit = static_cast<size_t>(n_atoms) * n_frames * dtype_size(dtype);
  MDX_TRY(to_device(x, location, unit * 3, st_x, stream, &dx));
  MDX_TRY(to_device(ke, location, unit, st_k, stream, &dk));
  MDX_TRY(to_device(pe, location, unit, st_p, stream, &dp));
  const bool streamed = stream_layout_ok(dtype, n_frames, {dx, dk, dp});
  const ChunkPlan plan =
      streamed ? plan_stream_chunks(n_atoms, n_frames) : plan_chunks(n_atoms, n_frames);
  MDX_TRY(partial.alloc(sizeof(double) * static_cast<size_t>(plan.n_chunks) * n_cols));
  dim3 grid(static_cast<unsigned>(plan.col_blocks), static_cast<unsigned>(plan.n_chunks));
  {
  KernelTimer timer(MDX_SLOT_FLUX, stream);
  if (streamed)
    hipLaunchKernelGGL(flux_stream_kernel<false>, grid, dim3(kThreads), 0, stream,
                       static_cast<const float *>(nullptr), static_cast<const float *>(dx),
                       static_cast<const float *>(dk), static_cast<const float *>(dp),
                       n_atoms, n_frames, plan.atoms_per_chunk, partial.as<double>());
  else if (dtype == MDX_F32)
    hipLaunchKernelGGL(energy_weighted_kernel<float>, grid, dim3(kThreads), 0, stream,
                       static_cast<const float *>(dx), static_cast<const float *>(dk),
                       static_cast<const float *>(dp), n_atoms, n_frames,
                       plan.atoms_per_chunk, partial.as<double>());
  else
    hipLaunchKernelGGL(energy_weighted_kernel<double>, grid, dim3(kThreads), 0, stream,
                       static_cast<const double *>(dx), static_cast<const double *>(dk),
                       static_cast<const double *>(dp), n_atoms, n_frames,
                       plan.atoms_per_chunk, partial.as<double>());
  }
  MDX_HIP(hipGetLastError());
  MDX_TRY(finish(partial, static_cast<int>(plan.n_chunks), 1, n_cols, out, out_location,
                 accumulate, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

int mdx_kinaci_heat_current(const void *pos, const void *vel, const void *force,
                            const void *pe, int dtype, int location, int64_t n_atoms,
                            int64_t n_frames, double dt, int have_carry, double *carry,
                            double *out, int out_location, int accumulate) {
  MDX_TRY(check_common(pos, dtype, location, n_atoms, n_frames, out, out_location));
  MDX_REQUIRE(vel != nullptr && force != nullptr && pe != nullptr && carry != nullptr,
              "null argument");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  const int64_t n_cols = n_frames * 3;
  if (n_atoms == 0) return zero_result(out, out_location, n_cols, accumulate, stream);
  if (!have_carry)
    for (int64_t a = 0; a < n_atoms; ++a) carry[a] = 0.0;
  if (n_frames == 0) return MDX_OK;
  DeviceBuffer st_x, st_v, st_f, st_p, partial, totals, d_carry, d_next;
  const void *dx = nullptr, *dv = nullptr, *df = nullptr, *dp = nullptr;
  const size_t unit = static_cast<size_t>(n_atoms) * n_frames * dtype_size(dtype);
  MDX_TRY(to_device(pos, location, unit * 3, st_x, stream, &dx));
  MDX_TRY(to_device(vel, location, unit * 3, st_v, stream, &dv));
  MDX_TRY(to_device(force, location, unit * 3, st_f, stream, &df));
  MDX_TRY(to_device(pe, location, unit, st_p, stream, &dp));
  const size_t carry_bytes = sizeof(double) * static_cast<size_t>(n_atoms);
  MDX_TRY(d_carry.alloc(carry_bytes));
  MDX_TRY(d_next.alloc(carry_bytes));
  MDX_HIP(hipMemcpyAsync(d_carry.ptr, carry, carry_bytes, hipMemcpyHostToDevice, stream));
  // float32, even frame count, aligned arrays: the streaming form (MDX_KINACI_SCAN=1 keeps
  // the two-launch scan, for comparison)
  const char *scan_env = getenv("MDX_KINACI_SCAN");
  if (stream_layout_ok(dtype, n_frames, {dx, dv, df, dp}) && (n_frames * 3) % 4 == 0 &&
      !(scan_env != nullptr && scan_env[0] == '1')) {
    const int64_t n_slices = ceil_div(n_frames, static_cast<int64_t>(kWaveFrames));
    const ChunkPlan plan_a = plan_stream_chunks(n_atoms, n_frames);
    const ChunkPlan plan_b = plan_chunks(n_atoms, n_cols / 4);
    const int chunks = static_cast<int>(plan_a.n_chunks + plan_b.n_chunks);
    MDX_TRY(partial.alloc(sizeof(double) * static_cast<size_t>(chunks) * n_cols));
    MDX_TRY(totals.alloc(carry_bytes * static_cast<size_t>(n_slices)));
    {
      KernelTimer timer(MDX_SLOT_FLUX, stream);
      hipLaunchKernelGGL(kinaci_stream_kernel,
                         dim3(static_cast<unsigned>(plan_a.col_blocks),
                              static_cast<unsigned>(plan_a.n_chunks)),
                         dim3(kThreads), 0, stream, static_cast<const float *>(dx),
                         static_cast<const float *>(dv), static_cast<const float *>(df),
                         static_cast<const float *>(dp), n_atoms, n_frames,
                         plan_a.atoms_per_chunk, dt, d_carry.as<double>(), totals.as<double>(),
                         partial.as<double>());
      hipLaunchKernelGGL(kinaci_offsets_kernel,
                         dim3(static_cast<unsigned>(ceil_div(n_atoms, 64))),
                         dim3(64), 0, stream, totals.as<double>(), n_slices, n_atoms, dt,
                         d_carry.as<double>(), d_next.as<double>());
      hipLaunchKernelGGL(sliced_weighted_sum_kernel,
                         dim3(static_cast<unsigned>(plan_b.col_blocks),
                              static_cast<unsigned>(plan_b.n_chunks)),
                         dim3(kThreads), 0, stream, static_cast<const float *>(dx),
                         totals.as<double>(), dt, n_atoms, n_cols,
                         static_cast<int64_t>(kWaveFrames) * 3, plan_b.atoms_per_chunk,
                         partial.as<double>() + plan_a.n_chunks * n_cols);
    }
    MDX_HIP(hipGetLastError());
    MDX_TRY(finish(partial, chunks, 1, n_cols, out, out_location, accumulate, stream));
    MDX_HIP(hipMemcpyAsync(carry, d_next.ptr, carry_bytes, hipMemcpyDeviceToHost, stream));
    MDX_HIP(hipStreamSynchronize(stream));
    return MDX_OK;
  }
  const ChunkPlan plan = plan_chunks(n_atoms, n_frames);
  MDX_TRY(partial.alloc(sizeof(double) * static_cast<size_t>(plan.n_chunks) * n_cols));
  MDX_TRY(totals.alloc(carry_bytes * static_cast<size_t>(plan.col_blocks)));
  dim3 grid(static_cast<unsigned>(plan.col_blocks), static_cast<unsigned>(plan.n_chunks));
  const unsigned atom_blocks = static_cast<unsigned>(ceil_div(n_atoms, 64));
#define MDX_KINACI(TYPE)                                                                      \
  do {                                                                                        \
    hipLaunchKernelGGL((kinaci_kernel<TYPE, false>), grid, dim3(kThreads), 0, stream,         \
                       static_cast<const TYPE *>(dx), static_cast<const TYPE *>(dv),          \
                       static_cast<const TYPE *>(df), static_cast<const TYPE *>(dp), n_atoms, \
                       n_frames, plan.atoms_per_chunk, dt, d_carry.as<double>(),              \
                       totals.as<double>(), partial.as<double>());                            \
    hipLaunchKernelGGL(kinaci_offsets_kernel, dim3(atom_blocks), dim3(64), 0, stream,         \
                       totals.as<double>(), plan.col_blocks, n_atoms, dt,                     \
                       d_carry.as<double>(), d_next.as<double>());                            \
    hipLaunchKernelGGL((kinaci_kernel<TYPE, true>), grid, dim3(kThreads), 0, stream,          \
                       static_cast<const TYPE *>(dx), static_cast<const TYPE *>(dv),          \
                       static_cast<const TYPE *>(df), static_cast<const TYPE *>(dp), n_atoms, \
                       n_frames, plan.atoms_per_chunk, dt, d_carry.as<double>(),              \
                       totals.as<double>(), partial.as<double>());                            \
  } while (0)
  {
    KernelTimer timer(MDX_SLOT_FLUX, stream);
    if (dtype == MDX_F32) MDX_KINACI(float); else MDX_KINACI(double);
  }
#undef MDX_KINACI
  MDX_HIP(hipGetLastError());
  MDX_TRY(finish(partial, static_cast<int>(plan.n_chunks), 1, n_cols, out, out_location,
                 accumulate, stream));
  MDX_HIP(hipMemcpyAsync(carry, d_next.ptr, carry_bytes, hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

}  // extern "C"
