// Angular distribution function: weighted histograms of the angle at atom i between
// all ordered pairs of its neighbours (j, k), j != k, both inside the cutoff.
//
// Three passes per block of frames:
//   adf_gather_kernel     species arrays [atoms][frames][3] -> X[frame][component][atom]
//   adf_neighbour_kernel  thread = atom i, loop over every j: literal float32 minimum
//                         image, distance, cutoff test in float16 as the reference
//                         does; neighbours (index, unit vector, length) go into
//                         per-atom lists sized from the mean density, a second run
//                         with the exact size follows only if a list overflowed;
//                         grid.z splits the j loop so that few frames fill the chip
//   adf_triplet_kernel    wave = atom i, its list staged in LDS, lanes stride over the
//                         unordered neighbour pairs (each stands for the reference's two
//                         ordered ones): angle = acos(clamp(u_ij . u_ik)),
//                         weight = 1 / (|r_ij| |r_ik|)^p, float64 LDS histograms per
//                         species triple, flushed once
// Larger systems replace the all-pairs loop by a cell list (see "cell lists" below).
//
// The all-pairs search is the same N^2 loop as the radial distribution function; with
// cutoffs of a few Angstrom a frame has ~N * n^2 triplets (n ~ 30 neighbours), far
// less work.  Compiled with -ffp-contract=off and correctly rounded divide / sqrt,
// so the vectors, lengths and unit vectors are bit-identical to a float32 NumPy
// evaluation; acosf / powf may differ from other libraries in the last place.
#include "mdx_common.h"

using namespace mdx;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxAdfSpecies = 8;
constexpr int kCountCopies = 32;

struct AdfBox {
  float box[3];
};

template <typename T>
__global__ __launch_bounds__(kThreads) void adf_gather_kernel(
    const void *const *__restrict__ species_ptr, const int32_t *__restrict__ atom_species,
    const int32_t *__restrict__ atom_local, const int64_t *__restrict__ frames, int64_t f0,
    int64_t frame_stride, int nf, int64_t n_atoms, int64_t n_pad, float *__restrict__ X) {
  // n_pad = n_atoms rounded up to a multiple of 4; the padding slots hold NaN, which
  // is never inside a cutoff
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= n_pad * nf) return;
  const int64_t a = e % n_pad;
  const int64_t fl = e / n_pad;
  float *dst = X + fl * 3 * n_pad + a;
  if (a >= n_atoms) {
    const float nan = __builtin_nanf("");
    dst[0] = nan;
    dst[n_pad] = nan;
    dst[2 * n_pad] = nan;
    return;
  }
  const int64_t frame = frames ? frames[fl] : f0 + fl;
  const T *src = static_cast<const T *>(species_ptr[atom_species[a]]) +
                 (static_cast<int64_t>(atom_local[a]) * frame_stride + frame) * 3;
  dst[0] = static_cast<float>(src[0]);
  dst[n_pad] = static_cast<float>(src[1]);
  dst[2 * n_pad] = static_cast<float>(src[2]);
}

// A block covers 256 atoms i and a chunk of atoms j (grid.z); counts[frame][i] must be
// zero at launch.  Every neighbour found raises counts[frame][i] and takes the next free
// slot of nbr_index / nbr_vec [frame][i][capacity] (vec = unit vector, length) if there
// is one; the host checks the largest count against the capacity afterwards.  The order of
// a list varies from run to run.
__global__ __launch_bounds__(kThreads) void adf_neighbour_kernel(
    const float *__restrict__ X, int64_t n_atoms, int64_t n_pad, AdfBox p, float cutoff,
    int capacity, int64_t j_chunk, int32_t *__restrict__ counts,
    int32_t *__restrict__ nbr_index, float4 *__restrict__ nbr_vec) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t f = blockIdx.y;
  const float *x = X + f * 3 * n_pad;
  const float *y = x + n_pad;
  const float *z = y + n_pad;
  const bool active = i < n_atoms;
  const float xi = active ? x[i] : 0.0f;
  const float yi = active ? y[i] : 0.0f;
  const float zi = active ? z[i] : 0.0f;
  const _Float16 cut16 = static_cast<_Float16>(cutoff);
  const float inv_box[3] = {1.0f / p.box[0], 1.0f / p.box[1], 1.0f / p.box[2]};
  const float s_far = (cutoff * 1.002f) * (cutoff * 1.002f);
  const int64_t base = (f * n_atoms + i) * capacity;
  // j_chunk is a multiple of 4: four j atoms per wave-uniform 16-byte load
  const int64_t j_begin = static_cast<int64_t>(blockIdx.z) * j_chunk;
  const int64_t j_end = min(n_pad, j_begin + j_chunk);
  for (int64_t j0 = j_begin; j0 < j_end; j0 += 4) {
    const float4 xs = *reinterpret_cast<const float4 *>(x + j0);
    const float4 ys = *reinterpret_cast<const float4 *>(y + j0);
    const float4 zs = *reinterpret_cast<const float4 *>(z + j0);
    const float xj[4] = {xs.x, xs.y, xs.z, xs.w};
    const float yj[4] = {ys.x, ys.y, ys.z, ys.w};
    const float zj[4] = {zs.x, zs.y, zs.z, zs.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float rx = xi - xj[u];
      float ry = yi - yj[u];
      float rz = zi - zj[u];
      // literal rint(r / box), without the divide unless the quotient is close to a tie
      rx = rx - exact_rint_quotient(rx, p.box[0], inv_box[0]) * p.box[0];
      ry = ry - exact_rint_quotient(ry, p.box[1], inv_box[1]) * p.box[1];
      rz = rz - exact_rint_quotient(rz, p.box[2], inv_box[2]) * p.box[2];
      const float s = (rx * rx + ry * ry) + rz * rz;
      // far pairs leave here, usually the whole wave at once: float16 rounding can
      // admit distances up to cutoff * (1 + 2^-11) only; NaN (padding) fails the
      // comparison too
      const bool near = s < s_far;
      if (__builtin_amdgcn_ballot_w64(near) == 0) continue;
      if (!near) continue;
      const float d = sqrtf(s);
      const _Float16 d16 = static_cast<_Float16>(d);
      // a zero distance (the atom itself) counts as the cutoff, i.e. outside; NaN
      // (padding) fails the comparison
      const bool inside = active && (d16 != static_cast<_Float16>(0.0f)) && (d16 < cut16);
      if (inside) {
        const int slot = atomicAdd(&counts[f * n_atoms + i], 1);
        if (slot < capacity) {
          nbr_index[base + slot] = static_cast<int32_t>(j0 + u);
          nbr_vec[base + slot] = make_float4(rx / d, ry / d, rz / d, d);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ cell lists
//
// For systems of a few thousand atoms and more, and cutoffs below a third of the box,
// the pair search runs over a grid of cells whose edges are at least 1.01 cutoffs: an
// atom's neighbours sit in its own and the 26 surrounding cells (periodic).  Every pair
// it finds goes through the same arithmetic as in adf_neighbour_kernel, and every pair
// that kernel would accept lies in those cells, so the lists hold the same neighbours.

struct AdfCells {
  int nc[3];
  int n_cells;
};

__device__ __forceinline__ int cell_coordinate(float x, float box, int nc) {
  float u = x / box;
  u = u - floorf(u);  // position folded into [0, 1]
  const int c = static_cast<int>(u * static_cast<float>(nc));
  return min(max(c, 0), nc - 1);
}

// cell_of[f][i], rank_in_cell[f][i] (arrival order) and cell_count[f][cell]; the counts
// must be zero at launch.
__global__ __launch_bounds__(kThreads) void adf_cell_assign_kernel(
    const float *__restrict__ X, int64_t n_atoms, int64_t n_pad, int nf, AdfBox p, AdfCells g,
    int32_t *__restrict__ cell_of, int32_t *__restrict__ rank_in_cell,
    int32_t *__restrict__ cell_count) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= n_atoms * nf) return;
  const int64_t f = e / n_atoms;
  const int64_t i = e - f * n_atoms;
  const float *x = X + f * 3 * n_pad;
  const int cx = cell_coordinate(x[i], p.box[0], g.nc[0]);
  const int cy = cell_coordinate(x[n_pad + i], p.box[1], g.nc[1]);
  const int cz = cell_coordinate(x[2 * n_pad + i], p.box[2], g.nc[2]);
  const int c = (cx * g.nc[1] + cy) * g.nc[2] + cz;
  cell_of[e] = c;
  rank_in_cell[e] = atomicAdd(&cell_count[f * g.n_cells + c], 1);
}

// cell_start[f][0 .. n_cells] = exclusive prefix sums of cell_count[f][.]; one block
// per frame, a thread sums a contiguous run of cells.
__global__ __launch_bounds__(kThreads) void adf_cell_scan_kernel(
    const int32_t *__restrict__ cell_count, int n_cells, int32_t *__restrict__ cell_start) {
  __shared__ int32_t run_total[kThreads];
  const int64_t f = blockIdx.x;
  const int32_t *count = cell_count + f * n_cells;
  int32_t *start = cell_start + f * (n_cells + 1);
  const int per = (n_cells + kThreads - 1) / kThreads;
  const int c0 = min(n_cells, static_cast<int>(threadIdx.x) * per);
  const int c1 = min(n_cells, c0 + per);
  int32_t sum = 0;
  for (int c = c0; c < c1; ++c) sum += count[c];
  run_total[threadIdx.x] = sum;
  __syncthreads();
  int32_t before = 0;
  for (int t = 0; t < static_cast<int>(threadIdx.x); ++t) before += run_total[t];
  for (int c = c0; c < c1; ++c) {
    start[c] = before;
    before += count[c];
  }
  if (threadIdx.x == kThreads - 1) start[n_cells] = before;
}

// sorted[f][slot] = (x, y, z, atom index as bits), atoms of one cell adjacent.
__global__ __launch_bounds__(kThreads) void adf_cell_scatter_kernel(
    const float *__restrict__ X, int64_t n_atoms, int64_t n_pad, int nf, int n_cells,
    const int32_t *__restrict__ cell_of, const int32_t *__restrict__ rank_in_cell,
    const int32_t *__restrict__ cell_start, float4 *__restrict__ sorted) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= n_atoms * nf) return;
  const int64_t f = e / n_atoms;
  const int64_t i = e - f * n_atoms;
  const float *x = X + f * 3 * n_pad;
  const int slot = cell_start[f * (n_cells + 1) + cell_of[e]] + rank_in_cell[e];
  sorted[f * n_atoms + slot] =
      make_float4(x[i], x[n_pad + i], x[2 * n_pad + i], __int_as_float(static_cast<int>(i)));
}

// Thread = one atom (in cell order), loop over the atoms of its 27 cells.  Outputs as
// adf_neighbour_kernel, except that counts need not be zeroed.
__global__ __launch_bounds__(kThreads) void adf_neighbour_cells_kernel(
    const float4 *__restrict__ sorted, int64_t n_atoms, AdfBox p, AdfCells g, float cutoff,
    int capacity, const int32_t *__restrict__ cell_of, const int32_t *__restrict__ cell_start,
    int32_t *__restrict__ counts, int32_t *__restrict__ nbr_index,
    float4 *__restrict__ nbr_vec) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (s >= n_atoms) return;
  const int64_t f = blockIdx.y;
  const float4 *atoms = sorted + f * n_atoms;
  const int32_t *start = cell_start + f * (g.n_cells + 1);
  const float4 me = atoms[s];
  const int i = __float_as_int(me.w);
  const int c = cell_of[f * n_atoms + i];
  const int cz = c % g.nc[2];
  const int cy = (c / g.nc[2]) % g.nc[1];
  const int cx = c / (g.nc[2] * g.nc[1]);
  const _Float16 cut16 = static_cast<_Float16>(cutoff);
  const float inv_box[3] = {1.0f / p.box[0], 1.0f / p.box[1], 1.0f / p.box[2]};
  const float s_far = (cutoff * 1.002f) * (cutoff * 1.002f);
  const int64_t base = (f * n_atoms + i) * capacity;
  int n = 0;
  for (int dx = -1; dx <= 1; ++dx) {
    const int nx = (cx + dx + g.nc[0]) % g.nc[0];
    for (int dy = -1; dy <= 1; ++dy) {
      const int ny = (cy + dy + g.nc[1]) % g.nc[1];
      for (int dz = -1; dz <= 1; ++dz) {
        const int nz = (cz + dz + g.nc[2]) % g.nc[2];
        const int cell = (nx * g.nc[1] + ny) * g.nc[2] + nz;
        const int j_end = start[cell + 1];
        for (int j = start[cell]; j < j_end; ++j) {
          const float4 other = atoms[j];
          float rx = me.x - other.x;
          float ry = me.y - other.y;
          float rz = me.z - other.z;
          rx = rx - exact_rint_quotient(rx, p.box[0], inv_box[0]) * p.box[0];
          ry = ry - exact_rint_quotient(ry, p.box[1], inv_box[1]) * p.box[1];
          rz = rz - exact_rint_quotient(rz, p.box[2], inv_box[2]) * p.box[2];
          const float sq = (rx * rx + ry * ry) + rz * rz;
          if (!(sq < s_far)) continue;
          const float d = sqrtf(sq);
          const _Float16 d16 = static_cast<_Float16>(d);
          if ((d16 != static_cast<_Float16>(0.0f)) && (d16 < cut16)) {
            if (n < capacity) {
              nbr_index[base + n] = __float_as_int(other.w);
              nbr_vec[base + n] = make_float4(rx / d, ry / d, rz / d, d);
            }
            ++n;
          }
        }
      }
    }
  }
  counts[f * n_atoms + i] = n;
}

// x^p: repeated multiplication for the integer powers 0..8 (within 2 ulp of powf, and
// far cheaper), powf otherwise.
__device__ __forceinline__ float small_power(float x, float p, int int_power) {
  if (int_power < 0) return powf(x, p);
  float r = 1.0f;
  float b = x;
  int e = int_power;
  while (e > 0) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}

// hist[combo][bin] (float64) and triples[combo] accumulate over the launch.
// LDS: double hist[n_combos * nbins] | double edges[nbins + 1] |
//      uint32 count[n_combos][kCountCopies] (one copy per lane pair: the lanes of a wave
//      mostly count the same species triple) |
//      per wave: float4 vec[stage] | int32 species[stage] (kStaged only)
//
// The reference walks over ordered neighbour pairs (j, k) and keeps those whose species
// triple (i, j, k) is in non-decreasing order.  (j, k) and (k, j) share angle and weight,
// so a wave evaluates every unordered pair once — enumerated as (a, a + s mod n), s = 1 ..
// n/2, which needs no square root — and books it once if only one order qualifies, twice if
// both do (equal species).  kStaged: the atom's list is first copied into LDS (it is read
// n times); lists longer than the staging area are read from global memory instead.
template <bool kStaged>
__global__ __launch_bounds__(kThreads) void adf_triplet_kernel(
    const int32_t *__restrict__ counts, const int32_t *__restrict__ nbr_index,
    const float4 *__restrict__ nbr_vec, const int32_t *__restrict__ atom_species,
    const int32_t *__restrict__ combo_of, int n_species, int n_combos, int64_t n_atoms,
    int capacity, int stage, int atoms_per_block, int nbins,
    const double *__restrict__ g_edges, float norm_power, double *__restrict__ g_hist,
    unsigned long long *__restrict__ g_triples) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *hist = reinterpret_cast<double *>(lds_raw);
  double *edges = hist + static_cast<size_t>(n_combos) * nbins;
  uint32_t *count = reinterpret_cast<uint32_t *>(edges + nbins + 1);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // staging area behind the counters, 16-byte aligned
  const size_t head = (sizeof(double) * (static_cast<size_t>(n_combos) * nbins + nbins + 1) +
                       sizeof(uint32_t) * n_combos * kCountCopies + 15) / 16 * 16;
  float4 *my_vec = reinterpret_cast<float4 *>(lds_raw + head) + static_cast<size_t>(wave) * stage;
  int32_t *my_species =
      reinterpret_cast<int32_t *>(reinterpret_cast<float4 *>(lds_raw + head) +
                                  static_cast<size_t>(kThreads / 64) * stage) +
      static_cast<size_t>(wave) * stage;
  for (int b = threadIdx.x; b < n_combos * nbins; b += kThreads) hist[b] = 0.0;
  for (int b = threadIdx.x; b <= nbins; b += kThreads) edges[b] = g_edges[b];
  for (int b = threadIdx.x; b < n_combos * kCountCopies; b += kThreads) count[b] = 0u;
  __syncthreads();
  const int64_t f = blockIdx.y;
  const int64_t i_begin = static_cast<int64_t>(blockIdx.x) * atoms_per_block;
  const int64_t i_end = min(n_atoms, i_begin + atoms_per_block);
  const double first = edges[0];
  const double scale = static_cast<double>(nbins) / (edges[nbins] - edges[0]);
  const int int_power =
      (norm_power >= 0.0f && norm_power <= 8.0f && norm_power == rintf(norm_power))
          ? static_cast<int>(norm_power)
          : -1;
  for (int64_t i = i_begin + wave; i < i_end; i += kThreads / 64) {
    const int n = min(counts[f * n_atoms + i], capacity);
    if (n < 2) continue;
    const int64_t base = (f * n_atoms + i) * capacity;
    const int si = atom_species[i];
    if (kStaged) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the previous atom's list is no longer read
      for (int a = lane; a < n; a += 64) {
        my_vec[a] = nbr_vec[base + a];
        my_species[a] = atom_species[nbr_index[base + a]];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // unordered pairs: n * full_steps of them with every a, then (n even) n/2 more
    const int full_steps = (n - 1) / 2;
    const int n_pairs = n * full_steps + ((n & 1) ? 0 : n / 2);
    const uint32_t magic = 0xFFFFFFFFu / static_cast<uint32_t>(n) + 1u;
    for (int pidx = lane; pidx < n_pairs; pidx += 64) {
      int a, step;
      if (pidx < n * full_steps) {
        // the multiply-high quotient is exact while pidx * n < 2^32
        step = (n < 1024) ? static_cast<int>(__umulhi(static_cast<uint32_t>(pidx), magic))
                          : pidx / n;
        a = pidx - step * n;
        step += 1;
      } else {
        a = pidx - n * full_steps;
        step = n / 2;
      }
      int b = a + step;
      if (b >= n) b -= n;
      float4 va, vb;
      int sj, sk;
      if (kStaged) {
        va = my_vec[a];
        vb = my_vec[b];
        sj = my_species[a];
        sk = my_species[b];
      } else {
        va = nbr_vec[base + a];
        vb = nbr_vec[base + b];
        sj = atom_species[nbr_index[base + a]];
        sk = atom_species[nbr_index[base + b]];
      }
      // (i, j, k) in non-decreasing species order, for either order of the pair
      const int lo = min(sj, sk), hi = max(sj, sk);
      const int combo = combo_of[(si * n_species + lo) * n_species + hi];
      if (combo < 0) continue;
      const uint32_t times = (sj == sk) ? 2u : 1u;
      float dot = (va.x * vb.x + va.y * vb.y) + va.z * vb.z;
      dot = fminf(fmaxf(dot, -1.0f), 1.0f);
      const float angle = acosf(dot);
      const float weight = 1.0f / small_power(va.w * vb.w, norm_power, int_power);
      const double v = static_cast<double>(angle);
      if (!(v >= first) || !(v <= edges[nbins])) continue;  // outside the range (or NaN)
      int bin = static_cast<int>((v - first) * scale);
      bin = max(0, min(bin, nbins - 1));
      while (bin > 0 && v < edges[bin]) --bin;
      while (bin < nbins - 1 && v >= edges[bin + 1]) ++bin;
      atomicAdd(&hist[combo * nbins + bin], static_cast<double>(weight) * times);
      atomicAdd(&count[combo * kCountCopies + (lane & (kCountCopies - 1))], times);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < n_combos * nbins; b += kThreads) {
    const double v = hist[b];
    if (v != 0.0) atomicAdd(&g_hist[b], v);
  }
  for (int b = threadIdx.x; b < n_combos; b += kThreads) {
    unsigned long long total = 0;
    for (int c = 0; c < kCountCopies; ++c) total += count[b * kCountCopies + c];
    if (total != 0) atomicAdd(&g_triples[b], total);
  }
}

__global__ void adf_max_kernel(const int32_t *__restrict__ counts, int64_t n,
                               int32_t *__restrict__ out) {
  int m = 0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x)
    m = max(m, counts[e]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

}  // namespace

extern "C" {

int mdx_adf_num_combinations(int n_species) {
  return n_species * (n_species + 1) * (n_species + 2) / 6;
}

int mdx_adf_accumulate(const void *const *species_pos, int dtype, int location, int n_species,
                       const int64_t *n_per_species, int64_t frame_stride,
                       const int64_t *frames, int64_t n_frames, const float box[3],
                       float cutoff, int nbins, const double *bin_edges, float norm_power,
                       double *hist, uint64_t *triple_counts) {
  MDX_REQUIRE(species_pos != nullptr && n_per_species != nullptr && box != nullptr &&
                  bin_edges != nullptr && hist != nullptr && triple_counts != nullptr,
              "null argument");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  MDX_REQUIRE(n_species >= 1 && n_species <= kMaxAdfSpecies, "n_species out of range (1..%d)",
              kMaxAdfSpecies);
  MDX_REQUIRE(nbins >= 1 && cutoff > 0.0f, "nbins and cutoff must be positive");
  MDX_REQUIRE(n_frames >= 0 && frame_stride >= 0, "negative frame count");
  for (int c = 0; c < 3; ++c) MDX_REQUIRE(box[c] > 0.0f, "box lengths must be positive");
  for (int b = 0; b < nbins; ++b)
    MDX_REQUIRE(bin_edges[b] < bin_edges[b + 1], "bin edges must increase");
  int64_t n_atoms = 0;
  for (int s = 0; s < n_species; ++s) {
    MDX_REQUIRE(n_per_species[s] >= 0, "negative species size");
    n_atoms += n_per_species[s];
  }
  MDX_REQUIRE(n_atoms < (int64_t(1) << 31), "too many atoms");
  if (frames != nullptr) {
    for (int64_t i = 0; i < n_frames; ++i)
      MDX_REQUIRE(frames[i] >= 0 && frames[i] < frame_stride, "frame index out of range");
  } else {
    MDX_REQUIRE(n_frames <= frame_stride, "n_frames exceeds frame_stride");
  }
  if (n_frames == 0 || n_atoms == 0) return MDX_OK;

  const int n_combos = mdx_adf_num_combinations(n_species);
  const size_t lds = sizeof(double) * (static_cast<size_t>(n_combos) * nbins + nbins + 1) +
                     sizeof(uint32_t) * n_combos * kCountCopies + 16;
  MDX_REQUIRE(lds <= 160 * 1024 - 512,
              "%d species triples x %d bins need %zu bytes of LDS per workgroup", n_combos,
              nbins, lds);

  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  // atom tables and the species-triple lookup (combinations with replacement, in order)
  std::vector<int32_t> atom_species(n_atoms), atom_local(n_atoms);
  {
    int64_t a = 0;
    for (int s = 0; s < n_species; ++s)
      for (int64_t l = 0; l < n_per_species[s]; ++l, ++a) {
        atom_species[a] = s;
        atom_local[a] = static_cast<int32_t>(l);
      }
  }
  std::vector<int32_t> combo_of(static_cast<size_t>(n_species) * n_species * n_species, -1);
  {
    int id = 0;
    for (int a = 0; a < n_species; ++a)
      for (int b = a; b < n_species; ++b)
        for (int c = b; c < n_species; ++c)
          combo_of[(static_cast<size_t>(a) * n_species + b) * n_species + c] = id++;
  }
  const size_t elem = dtype_size(dtype);
  std::vector<DeviceBuffer> staged(location == MDX_HOST ? n_species : 0);
  std::vector<const void *> ptrs(n_species, nullptr);
  for (int s = 0; s < n_species; ++s) {
    if (n_per_species[s] == 0) continue;
    MDX_REQUIRE(species_pos[s] != nullptr, "positions of species %d are null", s);
    if (location == MDX_HOST) {
      MDX_TRY(to_device(species_pos[s], MDX_HOST,
                        static_cast<size_t>(n_per_species[s]) * frame_stride * 3 * elem,
                        staged[s], stream, &ptrs[s]));
    } else {
      ptrs[s] = species_pos[s];
    }
  }
  DeviceBuffer d_ptrs, d_species, d_local, d_combo, d_frames, d_edges, d_hist, d_triples;
  MDX_TRY(d_ptrs.alloc(sizeof(void *) * n_species));
  MDX_TRY(d_species.alloc(sizeof(int32_t) * n_atoms));
  MDX_TRY(d_local.alloc(sizeof(int32_t) * n_atoms));
  MDX_TRY(d_combo.alloc(sizeof(int32_t) * combo_of.size()));
  MDX_TRY(d_edges.alloc(sizeof(double) * (nbins + 1)));
  MDX_TRY(d_hist.alloc(sizeof(double) * n_combos * nbins));
  MDX_TRY(d_triples.alloc(sizeof(unsigned long long) * n_combos));
  MDX_HIP(hipMemcpyAsync(d_ptrs.ptr, ptrs.data(), sizeof(void *) * n_species,
                         hipMemcpyHostToDevice, stream));
  MDX_HIP(hipMemcpyAsync(d_species.ptr, atom_species.data(), sizeof(int32_t) * n_atoms,
                         hipMemcpyHostToDevice, stream));
  MDX_HIP(hipMemcpyAsync(d_local.ptr, atom_local.data(), sizeof(int32_t) * n_atoms,
                         hipMemcpyHostToDevice, stream));
  MDX_HIP(hipMemcpyAsync(d_combo.ptr, combo_of.data(), sizeof(int32_t) * combo_of.size(),
                         hipMemcpyHostToDevice, stream));
  MDX_HIP(hipMemcpyAsync(d_edges.ptr, bin_edges, sizeof(double) * (nbins + 1),
                         hipMemcpyHostToDevice, stream));
  MDX_HIP(hipMemsetAsync(d_hist.ptr, 0, sizeof(double) * n_combos * nbins, stream));
  MDX_HIP(hipMemsetAsync(d_triples.ptr, 0, sizeof(unsigned long long) * n_combos, stream));
  if (frames != nullptr) {
    MDX_TRY(d_frames.alloc(sizeof(int64_t) * n_frames));
    MDX_HIP(hipMemcpyAsync(d_frames.ptr, frames, sizeof(int64_t) * n_frames,
                           hipMemcpyHostToDevice, stream));
  }

  // Neighbour lists are written in one pass into slots sized from the mean density
  // (twice the expected neighbourhood plus a margin); a frame block whose largest
  // neighbourhood does not fit is searched again with the exact size.
  const int64_t n_pad = (n_atoms + 3) / 4 * 4;
  const double reach = static_cast<double>(cutoff) * 1.002;
  const double expected = static_cast<double>(n_atoms) /
                          (static_cast<double>(box[0]) * box[1] * box[2]) * 4.18879020478639 *
                          reach * reach * reach;
  int64_t guess = static_cast<int64_t>(2.0 * expected) + 32;
  if (guess > n_atoms) guess = n_atoms;
  // frames per block of work: packed image bounded to ~256 MiB, lists to ~8 GiB
  int64_t chunk = (int64_t(256) << 20) / (n_pad * 16);
  const int64_t list_chunk = (int64_t(8) << 30) / (n_atoms * guess * 20);
  if (chunk > list_chunk) chunk = list_chunk;
  if (chunk < 1) chunk = 1;
  if (chunk > n_frames) chunk = n_frames;
  if (chunk > 65535) chunk = 65535;
  // cell grid: edges of at least 1.01 cutoffs, at least three cells along every axis
  // (otherwise a periodic neighbour cell would be visited twice), enough atoms to pay
  // for the sort
  AdfCells cells{{1, 1, 1}, 1};
  bool use_cells = true;
  for (int c = 0; c < 3; ++c) {
    const double fit = static_cast<double>(box[c]) / (static_cast<double>(cutoff) * 1.01);
    cells.nc[c] = fit > 128.0 ? 128 : static_cast<int>(fit);
    if (cells.nc[c] < 3) use_cells = false;
  }
  cells.n_cells = cells.nc[0] * cells.nc[1] * cells.nc[2];
  int64_t cell_min_atoms = 2048;
  if (const char *env = getenv("MDX_ADF_CELL_MIN_ATOMS")) cell_min_atoms = atoll(env);
  if (n_atoms < cell_min_atoms) use_cells = false;
  if (use_cells) {
    // count and start tables bounded to ~1 GiB per block of frames
    const int64_t cell_chunk = (int64_t(1) << 30) / (static_cast<int64_t>(cells.n_cells) * 8);
    if (chunk > cell_chunk) chunk = cell_chunk < 1 ? 1 : cell_chunk;
  }
  DeviceBuffer d_X, d_counts, d_max, d_index, d_vec;
  DeviceBuffer d_cell_of, d_rank, d_cell_count, d_cell_start, d_sorted;
  if (use_cells) {
    MDX_TRY(d_cell_of.alloc(sizeof(int32_t) * n_atoms * chunk));
    MDX_TRY(d_rank.alloc(sizeof(int32_t) * n_atoms * chunk));
    MDX_TRY(d_cell_count.alloc(sizeof(int32_t) * cells.n_cells * chunk));
    MDX_TRY(d_cell_start.alloc(sizeof(int32_t) * (cells.n_cells + 1) * chunk));
    MDX_TRY(d_sorted.alloc(sizeof(float4) * n_atoms * chunk));
  }
  MDX_TRY(d_X.alloc(sizeof(float) * 3 * n_pad * chunk));
  MDX_TRY(d_counts.alloc(sizeof(int32_t) * n_atoms * chunk));
  MDX_TRY(d_max.alloc(sizeof(int32_t)));
  AdfBox p{{box[0], box[1], box[2]}};
  const unsigned atom_blocks = static_cast<unsigned>(ceil_div(n_atoms, kThreads));
  for (int64_t f0 = 0; f0 < n_frames; f0 += chunk) {
    const int nf = static_cast<int>(std::min<int64_t>(chunk, n_frames - f0));
    const dim3 ggrid(static_cast<unsigned>(ceil_div(n_pad * nf, kThreads)));
    const int64_t *fl = frames ? d_frames.as<int64_t>() + f0 : nullptr;
    if (dtype == MDX_F32) {
      hipLaunchKernelGGL(adf_gather_kernel<float>, ggrid, dim3(kThreads), 0, stream,
                         d_ptrs.as<const void *>(), d_species.as<int32_t>(),
                         d_local.as<int32_t>(), fl, f0, frame_stride, nf, n_atoms, n_pad,
                         d_X.as<float>());
    } else {
      hipLaunchKernelGGL(adf_gather_kernel<double>, ggrid, dim3(kThreads), 0, stream,
                         d_ptrs.as<const void *>(), d_species.as<int32_t>(),
                         d_local.as<int32_t>(), fl, f0, frame_stride, nf, n_atoms, n_pad,
                         d_X.as<float>());
    }
    MDX_HIP(hipGetLastError());
    // j chunks: enough blocks to fill the chip
    int64_t j_chunks = ceil_div(2048, static_cast<int64_t>(atom_blocks) * nf);
    if (j_chunks > ceil_div(n_atoms, 256)) j_chunks = ceil_div(n_atoms, 256);
    if (j_chunks < 1) j_chunks = 1;
    const int64_t j_chunk = (ceil_div(n_pad, j_chunks) + 3) / 4 * 4;
    j_chunks = ceil_div(n_pad, j_chunk);
    const dim3 ngrid(atom_blocks, static_cast<unsigned>(nf), static_cast<unsigned>(j_chunks));
    const size_t count_bytes = sizeof(int32_t) * n_atoms * nf;
    if (use_cells) {
      const dim3 agrid(static_cast<unsigned>(ceil_div(n_atoms * nf, kThreads)));
      MDX_HIP(hipMemsetAsync(d_cell_count.ptr, 0, sizeof(int32_t) * cells.n_cells * nf, stream));
      hipLaunchKernelGGL(adf_cell_assign_kernel, agrid, dim3(kThreads), 0, stream,
                         d_X.as<float>(), n_atoms, n_pad, nf, p, cells, d_cell_of.as<int32_t>(),
                         d_rank.as<int32_t>(), d_cell_count.as<int32_t>());
      MDX_HIP(hipGetLastError());
      hipLaunchKernelGGL(adf_cell_scan_kernel, dim3(static_cast<unsigned>(nf)), dim3(kThreads),
                         0, stream, d_cell_count.as<int32_t>(), cells.n_cells,
                         d_cell_start.as<int32_t>());
      MDX_HIP(hipGetLastError());
      hipLaunchKernelGGL(adf_cell_scatter_kernel, agrid, dim3(kThreads), 0, stream,
                         d_X.as<float>(), n_atoms, n_pad, nf, cells.n_cells,
                         d_cell_of.as<int32_t>(), d_rank.as<int32_t>(),
                         d_cell_start.as<int32_t>(), d_sorted.as<float4>());
      MDX_HIP(hipGetLastError());
    }
    int32_t capacity = static_cast<int32_t>(guess);
    for (int attempt = 0; attempt < 2; ++attempt) {
      const size_t slots = static_cast<size_t>(n_atoms) * nf * capacity;
      MDX_REQUIRE(slots * 20 <= (size_t(64) << 30),
                  "neighbour lists need %zu bytes (largest neighbourhood %d atoms)", slots * 20,
                  capacity);
      MDX_TRY(d_index.reserve(sizeof(int32_t) * slots));
      MDX_TRY(d_vec.reserve(sizeof(float4) * slots));
      if (use_cells) {
        hipLaunchKernelGGL(adf_neighbour_cells_kernel,
                           dim3(atom_blocks, static_cast<unsigned>(nf)), dim3(kThreads), 0,
                           stream, d_sorted.as<float4>(), n_atoms, p, cells, cutoff, capacity,
                           d_cell_of.as<int32_t>(), d_cell_start.as<int32_t>(),
                           d_counts.as<int32_t>(), d_index.as<int32_t>(), d_vec.as<float4>());
      } else {
        MDX_HIP(hipMemsetAsync(d_counts.ptr, 0, count_bytes, stream));
        hipLaunchKernelGGL(adf_neighbour_kernel, ngrid, dim3(kThreads), 0, stream,
                           d_X.as<float>(), n_atoms, n_pad, p, cutoff, capacity, j_chunk,
                           d_counts.as<int32_t>(), d_index.as<int32_t>(), d_vec.as<float4>());
      }
      MDX_HIP(hipGetLastError());
      MDX_HIP(hipMemsetAsync(d_max.ptr, 0, sizeof(int32_t), stream));
      hipLaunchKernelGGL(adf_max_kernel, dim3(256), dim3(kThreads), 0, stream,
                         d_counts.as<int32_t>(), n_atoms * nf, d_max.as<int32_t>());
      MDX_HIP(hipGetLastError());
      int32_t largest = 0;
      MDX_HIP(hipMemcpyAsync(&largest, d_max.ptr, sizeof(int32_t), hipMemcpyDeviceToHost,
                             stream));
      MDX_HIP(hipStreamSynchronize(stream));
      if (largest <= capacity) {
        if (largest < 2) capacity = 0;  // no atom has two neighbours: no triplets
        break;
      }
      MDX_REQUIRE(attempt == 0, "neighbour count changed between two searches of a frame block");
      capacity = largest;
    }
    if (capacity < 2) continue;
    const int atoms_per_block = 64;
    const dim3 tgrid(static_cast<unsigned>(ceil_div(n_atoms, atoms_per_block)),
                     static_cast<unsigned>(nf));
    // staging area of the neighbour lists: one list per wave if LDS has the room
    const size_t lds_head = (lds + 15) / 16 * 16;
    const size_t lds_staged = lds_head + static_cast<size_t>(kThreads / 64) * capacity * 20;
    const bool staged = lds_staged <= 96 * 1024;
    const size_t lds_launch = staged ? lds_staged : lds_head;
    if (staged) {
      MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(adf_triplet_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds_launch)));
      hipLaunchKernelGGL(adf_triplet_kernel<true>, tgrid, dim3(kThreads), lds_launch, stream,
                         d_counts.as<int32_t>(), d_index.as<int32_t>(), d_vec.as<float4>(),
                         d_species.as<int32_t>(), d_combo.as<int32_t>(), n_species, n_combos,
                         n_atoms, capacity, capacity, atoms_per_block, nbins,
                         d_edges.as<double>(), norm_power, d_hist.as<double>(),
                         d_triples.as<unsigned long long>());
    } else {
      MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(adf_triplet_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds_launch)));
      hipLaunchKernelGGL(adf_triplet_kernel<false>, tgrid, dim3(kThreads), lds_launch, stream,
                         d_counts.as<int32_t>(), d_index.as<int32_t>(), d_vec.as<float4>(),
                         d_species.as<int32_t>(), d_combo.as<int32_t>(), n_species, n_combos,
                         n_atoms, capacity, 0, atoms_per_block, nbins, d_edges.as<double>(),
                         norm_power, d_hist.as<double>(), d_triples.as<unsigned long long>());
    }
    MDX_HIP(hipGetLastError());
  }
  std::vector<double> h_hist(static_cast<size_t>(n_combos) * nbins);
  std::vector<unsigned long long> h_triples(n_combos);
  MDX_HIP(hipMemcpyAsync(h_hist.data(), d_hist.ptr, sizeof(double) * h_hist.size(),
                         hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipMemcpyAsync(h_triples.data(), d_triples.ptr,
                         sizeof(unsigned long long) * n_combos, hipMemcpyDeviceToHost,
                         stream));
  MDX_HIP(hipStreamSynchronize(stream));
  for (size_t b = 0; b < h_hist.size(); ++b) hist[b] += h_hist[b];
  for (int c = 0; c < n_combos; ++c) triple_counts[c] += h_triples[c];
  return MDX_OK;
}

}  // extern "C"
