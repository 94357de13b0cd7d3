// Radial distribution function: pair-distance histogramming on gfx950.
//
// Data flow per call of mdx_rdf_accumulate:
//   1. pack kernel: the reference layout [atoms][frames][3] (one array per
//      species) is gathered into a frame-major structure-of-arrays image
//      X[frame][component][slot] whose slots are grouped in tiles of 256 atoms,
//      one species per tile, padded with NaN (a NaN coordinate can never pass the
//      cutoff test, so padding needs no bounds checks in the pair loop);
//   2. pair kernel.  Each lane keeps one i-atom in registers; the 256 j-atoms of the
//      tile are the same for every lane and come through wave-uniform scalar loads
//      straight from the packed image; the species-pair histogram lives in LDS
//      (ds_add_u32) and is flushed with 64-bit global atomics.
//        rdf_fast_kernel  (default): persistent blocks, each walking a contiguous,
//          cost-balanced range of (tile pair, frame) items and flushing its histogram
//          only when the species-pair key changes;
//        rdf_tile_kernel  : one block per (tile pair, frame group); the literal
//          kernel, the table kernel for frames the fast kernel leaves out.
//
// Sorted frames: every kernel evaluates all pairs of a frame, as the reference does.  For
// species of many tiles (mdx_rdf_create decides) the frames are first sorted in space --
// rdf_sort_kernel permutes the atoms of every species along a Hilbert curve of cells, frame
// by frame, and records the bounding boxes of every tile of 256 atoms and of its quarters --
// and the fast kernel (a) leaves out the minimum-image fold of a component wherever the
// boxes of the two 64-atom groups show that no difference exceeds half a box length, and
// (b) when the cutoff is short against the box, skips the frames of a tile pair (and the
// quarters of a j-tile) whose boxes are further apart, over the periodic images, than the
// cutoff plus a margin.  The pair set of a frame does not depend on the order of the atoms
// within a species, so the counts are unchanged.
//
// Three pair functions produce identical bins:
//   literal : the reference's float32 operations one by one (IEEE divide, rint,
//             separate multiply/subtract, (x*x+y*y)+z*z, IEEE sqrt, d < cutoff,
//             float64 divide by the bin width, min, truncation);
//   table   : every step of the literal sequence after the coordinate difference
//             is a monotone function of one float32 value, so it can be replaced
//             by comparisons against float32 thresholds that the host computes
//             with the literal operations (bisection over bit patterns):
//               * minimum image: rint(fl(|r|/L)) >= 1 exactly when |r| >= t_wrap.
//                 A per-frame flag (extent kernel) says whether every coordinate
//                 difference of the frame stays below the point where the rint
//                 would reach 2; such frames ("wrapped") use |r| - (|r| >= t_wrap
//                 ? L : 0), whose square equals the literal one.  Other frames use
//                 a reciprocal multiply and fall back to the IEEE divide whenever
//                 the quotient is too close to a rounding boundary;
//               * binning: bin(s) of the squared distance is described by one
//                 threshold per bin edge.  The kernel takes floor(v_sqrt_f32(s) *
//                 nbins/cutoff); only when that product lies within the error
//                 bound of an integer does it consult the two neighbouring
//                 thresholds (kept in LDS).
//   fast    : the table kernel still spends most of its instructions on reproducing
//             the literal roundings of every pair.  The fast loop drops that: it
//             evaluates an approximation of the bin position whose distance from the
//             literal one is bounded on the host (rdf_fast_setup), and only pairs
//             whose position lies within that bound of a bin edge (well under 1 %)
//             are evaluated again, literally, against the thresholds.  Those pairs
//             are noted in a queue of their wave in LDS and corrected in batches of 64,
//             so the hot loop has no divergent slow path.
//
// This file must be compiled with -ffp-contract=off: a fused multiply-add in the
// minimum-image or the squared-distance expression changes the rounding and
// therefore the bins.
#include <cmath>

#include <cstdlib>
#include <thread>

#include "mdx_common.h"

using namespace mdx;

namespace {

constexpr int kTile = 256;          // atoms per tile == threads per block
// Words behind the fast kernel's LDS histogram that take the pairs beyond the cutoff, one
// per lane of a wave, so that every pair is counted somewhere and none needs a branch.
constexpr int kDumpWords = 64;
constexpr int kMaxFramesPerGroup = 64;

struct TilePair {
  int32_t tile_i;
  int32_t tile_j;
  int32_t key;
  int32_t diagonal;  // 1: same tile on both sides, only j > i counts
  int32_t j_count;   // atoms in the j tile (the rest of its 256 slots is padding)
  // position of the pair's key in the (sharded) pair list, which is sorted by key
  int32_t key_first_pair;
  int32_t key_pairs;
  int32_t pad_;      // 32 bytes: an entry is fetched with two 16-byte scalar loads
};

struct RdfParams {
  float box[3];
  float inv_box[3];
  float t_wrap[3];     // smallest |r| whose literal rint(r / box) is >= 1
  float t_far[3];      // smallest |r| whose literal rint(r / box) is >= 2
  float cutoff;        // float32 cutoff (literal kernel)
  float s_cut;         // smallest float32 s with sqrtf(s) >= cutoff (table kernel)
  float inv_step;      // float32(nbins / cutoff), approximate bin scale
  float edge_eps;      // error bound of the approximate bin position
  float edge_eps2;     // twice that
  double step;         // double(cutoff - 0.0f) / double(nbins)
  int nbins;
  // fast kernel (rdf_fast_setup)
  float half_box[3];   // box / 2, exact
  float fast_magic;    // 2^k + shift: adding it leaves the bin position in the mantissa
  uint32_t fast_frac_bits;   // F = 23 - k fraction bits
  uint32_t fast_addr_mask;   // ((1 << k) - 1) << 2: byte offset of the bin's LDS word
  uint32_t fast_frac_mask;   // (1 << F) - 1
  uint32_t fast_edge_units;  // E: a fraction below E grid units means "near an edge"
  uint32_t fast_keep_bits;   // float bits below this: position < nbins + E grid units
  uint32_t fast_dump_bits;   // bits of position nbins + 1 with every fraction bit set
  int fast_ok;               // 0: the bound is too wide for this geometry, do not use
  float cull_gap2;           // tile pairs with a squared box gap above this hold no pair in reach
  int cull_on;               // sorted frames: skip what the boxes put out of reach
#ifdef MDX_TIMING_SWITCHES
  int timing_mask;           // >= 0: every quarter takes this loop variant (wrong counts)
#endif
};

// ---------------------------------------------------------------- pack kernel

// Gathers one block of frames into X[frame][component][slot].
// slot_species[slot] < 0 marks a padding slot.  Reads run along the frame axis
// (contiguous in the source when the frame list is contiguous), writes along the
// slot axis; the exchange goes through LDS.
template <typename T>
__global__ __launch_bounds__(256) void rdf_pack_kernel(
    const void *const *__restrict__ species_ptr, const int32_t *__restrict__ slot_species,
    const int32_t *__restrict__ slot_atom, const int64_t *__restrict__ frames,
    int64_t frame_base, int64_t frame_stride, int n_frames, int64_t n_pad,
    float *__restrict__ packed) {
  constexpr int kSlots = 64;
  constexpr int kFrames = 16;
  __shared__ float sh[kSlots][kFrames * 3 + 1];
  const int64_t slot0 = static_cast<int64_t>(blockIdx.x) * kSlots;
  const int f0 = blockIdx.y * kFrames;
  // read phase: (slot, frame, component) with frame*3+component fastest
  for (int e = threadIdx.x; e < kSlots * kFrames * 3; e += blockDim.x) {
    const int s = e / (kFrames * 3);
    const int r = e - s * (kFrames * 3);
    const int f = r / 3;
    const int c = r - f * 3;
    float v = __builtin_nanf("");
    const int64_t slot = slot0 + s;
    if (slot < n_pad && f0 + f < n_frames) {
      const int sp = slot_species[slot];
      if (sp >= 0) {
        const int64_t frame =
            frames ? frames[f0 + f] : (frame_base + f0 + f);
        const T *src = static_cast<const T *>(species_ptr[sp]);
        v = static_cast<float>(
            src[(static_cast<int64_t>(slot_atom[slot]) * frame_stride + frame) * 3 + c]);
      }
    }
    sh[s][r] = v;
  }
  __syncthreads();
  // write phase: slot fastest
  for (int e = threadIdx.x; e < kSlots * kFrames * 3; e += blockDim.x) {
    const int r = e / kSlots;
    const int s = e - r * kSlots;
    const int f = r / 3;
    const int c = r - f * 3;
    const int64_t slot = slot0 + s;
    if (slot < n_pad && f0 + f < n_frames) {
      packed[(static_cast<int64_t>(f0 + f) * 3 + c) * n_pad + slot] = sh[s][r];
    }
  }
}

// ---------------------------------------------------------------- sort kernel

struct SortSpecies {
  int32_t first_slot;  // first slot of the species in a packed frame
  int32_t used;        // atoms that take part
  int32_t tiles;       // tiles of 256 slots
  int32_t cell_bits;   // cells per axis = 1 << cell_bits (at most 4)
};

// Position of every cell along a Hilbert curve, for grids of 2, 4, 8 and 16 cells per axis
// back to back (8 + 64 + 512 + 4096 entries); entry = rank of cell (z * g + y) * g + x.
constexpr int kCurveEntries = 8 + 64 + 512 + 4096;
__host__ __device__ inline int curve_offset(int cell_bits) {
  return cell_bits == 1 ? 0 : cell_bits == 2 ? 8 : cell_bits == 3 ? 72 : 584;
}

// One block per (species, frame): counting sort of the species' atoms by cell, cells along
// a Hilbert curve (consecutive cells share a face, so 256 consecutive atoms stay together),
// then the bounding boxes of every tile of the sorted frame and of its four quarters of 64
// slots.  boxes: per frame [5][n_tiles][...]: n_tiles tile boxes, then 4 n_tiles quarter
// boxes, six floats each (lowest corner, highest corner); a quarter without atoms gets
// (+inf, -inf), which is out of everybody's reach.
// The sorted order is a function of the frame alone (within a cell: the order of the
// input), never of the timing of the block's threads: ranks that share the tile pairs of
// a frame sort it each for themselves, and their tiles must hold the same atoms.  Counts
// come from LDS atomics (whose order does not matter), the cells' first positions from a
// scan in curve order.  Placement, usual case (a species of at most kSortIndexed atoms, no
// cell above kSortCellMax): the atoms' indices are scattered into an LDS list with atomics,
// in whatever order, every cell's piece of the list is then sorted by index (one thread
// per cell, insertion sort of about ten entries), and the coordinates are gathered along
// the list.  Otherwise (large species, clustered atoms): the four waves take a quarter of
// the atoms each, in input order, with a table of cell counts per wave, and place 64
// atoms at a time with one LDS atomic each -- a lane alone in its cell has its position;
// lanes that share a cell are found with ballots and take the cell's positions lowest lane
// first.  Both give the same order.
// frame_lo holds the frame's smallest coordinate per component, so the cell grid follows a
// trajectory that is not wrapped into [0, L); what lies more than a box length above it
// (frames of class 1, see rdf_extent_kernel) goes into the last cells.  Frames of class 0
// are copied unsorted (they go to the table kernel, which skips nothing); their boxes are
// never read.
// Dynamic LDS: uint32 [5][n_cells] | uint16 list[used] (species of at most kSortIndexed).
constexpr int kSortIndexed = 12288;
constexpr uint32_t kSortCellMax = 64;
__global__ __launch_bounds__(256) void rdf_sort_kernel(
    const float *__restrict__ packed, float *__restrict__ sorted, int64_t n_pad, int n_tiles,
    const SortSpecies *__restrict__ species, const int32_t *__restrict__ frame_wrapped,
    const float *__restrict__ frame_lo, const uint16_t *__restrict__ curve, RdfParams p,
    float *__restrict__ boxes) {
  extern __shared__ __align__(16) uint32_t sort_lds[];
  __shared__ uint32_t wave_total[4];
  __shared__ uint32_t largest_cell;
  const SortSpecies sp = species[blockIdx.x];
  const int f = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *X = packed + static_cast<int64_t>(f) * 3 * n_pad + sp.first_slot;
  float *Y = sorted + static_cast<int64_t>(f) * 3 * n_pad + sp.first_slot;
  const int slots = sp.tiles * kTile;
  if (frame_wrapped[f] == 0) {
    for (int c = 0; c < 3; ++c)
      for (int i = tid; i < slots; i += 256) Y[c * n_pad + i] = X[c * n_pad + i];
    return;
  }
  const int n_cells = 1 << (3 * sp.cell_bits);
  const uint32_t per_axis = 1u << sp.cell_bits;
  uint32_t *cell_count = sort_lds;                        // all atoms of a cell
  uint32_t *own = sort_lds + (1 + wave) * n_cells;        // this wave's atoms of a cell
  for (int i = tid; i < 5 * n_cells; i += 256) sort_lds[i] = 0u;
  if (tid == 0) largest_cell = 0u;
  __syncthreads();
  const uint16_t *rank_of_cell = curve + curve_offset(sp.cell_bits);
  float lo[3], scale[3];
  for (int c = 0; c < 3; ++c) {
    lo[c] = frame_lo[3 * f + c];
    scale[c] = static_cast<float>(per_axis) / p.box[c];
  }
  auto cell_of = [&](int i) -> uint32_t {
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = (X[c * n_pad + i] - lo[c]) * scale[c];
      // NaN and negative values convert to 0; the top is clamped
      q[c] = min(static_cast<uint32_t>(fmaxf(t, 0.0f)), per_axis - 1u);
    }
    return rank_of_cell[(((q[2] << sp.cell_bits) | q[1]) << sp.cell_bits) | q[0]];
  };
  // the wave's atoms: a quarter of the species, in whole groups of 64
  const int per_wave = ((sp.used + 255) / 256) * 64;
  const int w_begin = min(sp.used, wave * per_wave);
  const int w_end = min(sp.used, w_begin + per_wave);
  for (int i = w_begin + lane; i < w_end; i += 64) atomicAdd(&own[cell_of(i)], 1u);
  __syncthreads();
  // exclusive scan over the cells' totals (16 cells per thread at most), then the first
  // position of every wave within the cell
  {
    const int per_thread = (n_cells + 255) / 256;
    const int c0 = tid * per_thread;
    uint32_t sum = 0, largest = 0;
    for (int k = 0; k < per_thread; ++k)
      if (c0 + k < n_cells) {
        uint32_t n = 0;
        for (int w = 0; w < 4; ++w) n += sort_lds[(1 + w) * n_cells + c0 + k];
        cell_count[c0 + k] = n;
        sum += n;
        largest = max(largest, n);
      }
    atomicMax(&largest_cell, largest);
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t below = __shfl_up(incl, off);
      if (lane >= off) incl += below;
    }
    if (lane == 63) wave_total[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_total[w];
    for (int k = 0; k < per_thread; ++k)
      if (c0 + k < n_cells) {
        for (int w = 0; w < 4; ++w) {
          const uint32_t n = sort_lds[(1 + w) * n_cells + c0 + k];
          sort_lds[(1 + w) * n_cells + c0 + k] = base;
          base += n;
        }
      }
  }
  __syncthreads();
  const int first_tile = sp.first_slot / kTile;
  float *frame_boxes = boxes + static_cast<int64_t>(f) * 5 * n_tiles * 6;
  float *quarter_boxes = frame_boxes + static_cast<int64_t>(n_tiles) * 6;
  const float nan = __builtin_nanf("");
  // the box of the 64 slots a wave holds (NaN padding is ignored by fminf / fmaxf)
  auto quarter_box = [&](int tile, const float v[3]) {
    float mn[3], mx[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = fminf(__builtin_inff(), v[c]);
      mx[c] = fmaxf(-__builtin_inff(), v[c]);
      for (int off = 32; off > 0; off >>= 1) {
        mn[c] = fminf(mn[c], __shfl_xor(mn[c], off));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off));
      }
    }
    if (lane == 0) {
      float *b = quarter_boxes + (static_cast<int64_t>(first_tile + tile) * 4 + wave) * 6;
      for (int c = 0; c < 3; ++c) {
        b[c] = mn[c];
        b[3 + c] = mx[c];
      }
    }
  };
  if (sp.used <= kSortIndexed && largest_cell <= kSortCellMax) {
    uint16_t *list = reinterpret_cast<uint16_t *>(sort_lds + 5 * n_cells);
    const uint32_t *cell_first = sort_lds + n_cells;  // wave 0's first position = the cell's
    for (int c = tid; c < n_cells; c += 256) cell_count[c] = cell_first[c];
    __syncthreads();
    for (int i = tid; i < sp.used; i += 256)
      list[atomicAdd(&cell_count[cell_of(i)], 1u)] = static_cast<uint16_t>(i);
    __syncthreads();
    for (int c = tid; c < n_cells; c += 256) {
      const int first = static_cast<int>(cell_first[c]), last = static_cast<int>(cell_count[c]);
      for (int a = first + 1; a < last; ++a) {
        const uint16_t v = list[a];
        int b = a;
        for (; b > first && list[b - 1] > v; --b) list[b] = list[b - 1];
        list[b] = v;
      }
    }
    __syncthreads();
    // gather along the list, tile by tile: wave w writes quarter w (padding included) and
    // takes its box from the values in hand
    for (int t = 0; t < sp.tiles; ++t) {
      const int pos = t * kTile + tid;
      float v[3] = {nan, nan, nan};
      if (pos < sp.used) {
        const int i = list[pos];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = X[c * n_pad + i];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) Y[c * n_pad + pos] = v[c];
      quarter_box(t, v);
    }
  } else {
    for (int i0 = w_begin; i0 < w_end; i0 += 64) {
      const int i = i0 + lane;
      const bool valid = i < w_end;
      const uint32_t cell = valid ? cell_of(i) : 0xFFFFFFFFu;
      // The wave's next free position in the lane's cell, before and after the group has
      // taken its places there (the wave's LDS operations keep their order).  A lane alone in
      // its cell -- the rule, with thousands of cells -- owns the one place; lanes that share
      // a cell are ranked by lane with ballots, lowest first.
      uint32_t first = 0u, taken = 0u;
      if (valid) {
        first = own[cell];
        atomicAdd(&own[cell], 1u);
        taken = own[cell] - first;
      }
      uint32_t dst = first;
      unsigned long long remaining = __ballot(taken > 1u);
#ifdef MDX_TIMING_SWITCHES
      if (p.timing_mask == 100 || p.timing_mask == 101) { remaining = 0ull; dst = i; }
#endif
      while (remaining != 0ull) {
        const int leader = __builtin_ctzll(remaining);
        const uint32_t c0 = static_cast<uint32_t>(
            __builtin_amdgcn_readlane(static_cast<int>(cell), leader));
        const unsigned long long same = __ballot(cell == c0);
        if (cell == c0)
          dst = first + static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
        remaining &= ~same;
      }
      if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Y[c * n_pad + dst] = X[c * n_pad + i];
      }
    }
    for (int i = sp.used + tid; i < slots; i += 256)
      for (int c = 0; c < 3; ++c) Y[c * n_pad + i] = nan;
#ifdef MDX_TIMING_SWITCHES
    if (p.timing_mask == 101) return;
#endif
    __syncthreads();  // the block reads back what it wrote
    for (int t = 0; t < sp.tiles; ++t) {
      float v[3];
      for (int c = 0; c < 3; ++c) v[c] = Y[c * n_pad + t * kTile + tid];
      quarter_box(t, v);
    }
  }
  __syncthreads();
  for (int t = tid; t < sp.tiles; t += 256) {
    const float *q = quarter_boxes + static_cast<int64_t>(first_tile + t) * 4 * 6;
    float *b = frame_boxes + static_cast<int64_t>(first_tile + t) * 6;
    for (int c = 0; c < 3; ++c) {
      b[c] = fminf(fminf(q[c], q[6 + c]), fminf(q[12 + c], q[18 + c]));
      b[3 + c] = fmaxf(fmaxf(q[3 + c], q[9 + c]), fmaxf(q[15 + c], q[21 + c]));
    }
  }
}

// Squared distance between two boxes (lo[3], hi[3]) over the periodic images of a frame
// whose coordinate differences stay below 1.5 box lengths: per component the smallest gap
// between the intervals when one of them is moved by -L, 0 or +L.  (A shift by two box
// lengths leaves at least half a box length, and the skipping that asks for this gap is only
// switched on for cutoffs below 0.45 of the shortest edge.)
__device__ __forceinline__ float box_gap2(const float *__restrict__ a,
                                          const float *__restrict__ b, const RdfParams &p) {
  float sum = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a0 = a[c], a1 = a[3 + c], b0 = b[c], b1 = b[3 + c], L = p.box[c];
    const float direct = fmaxf(0.0f, fmaxf(b0 - a1, a0 - b1));
    const float up = fmaxf(0.0f, fmaxf((b0 + L) - a1, a0 - (b1 + L)));
    const float down = fmaxf(0.0f, fmaxf((b0 - L) - a1, a0 - (b1 - L)));
    const float g = fminf(direct, fminf(up, down));
    sum += g * g;
  }
  return sum;
}

// ------------------------------------------------------------- pair functions

// The reference's sequence of float32 operations, literally.
__device__ __forceinline__ void pair_literal(float xi, float yi, float zi, float4 pj,
                                             const RdfParams &p, bool allowed,
                                             uint32_t *hist) {
  float rx = pj.x - xi;
  float ry = pj.y - yi;
  float rz = pj.z - zi;
  rx = rx - rintf(rx / p.box[0]) * p.box[0];
  ry = ry - rintf(ry / p.box[1]) * p.box[1];
  rz = rz - rintf(rz / p.box[2]) * p.box[2];
  const float s = (rx * rx + ry * ry) + rz * rz;
  const float d = sqrtf(s);
  if (allowed && d < p.cutoff) {
    double q = static_cast<double>(d) / p.step;
    q = fmin(q, static_cast<double>(p.nbins - 1));
    atomicAdd(&hist[static_cast<int>(q)], 1u);
  }
}

// Classifies every frame by the spread of its coordinates: 2 = "compact", every
// coordinate difference is at most one box length (what a trajectory wrapped into the box
// gives); 1 = "wrapped", the differences stay below t_far (about 1.5 box lengths), so the
// minimum-image shift is at most one box length (atoms that have left the box a little
// since they were last wrapped) -- the fast kernel takes both; 0 = neither.
// One block per frame; NaN padding is ignored by fminf/fmaxf.  frame_lo, if given, receives
// the frame's smallest coordinate per component.
__global__ __launch_bounds__(256) void rdf_extent_kernel(const float *__restrict__ packed,
                                                         int64_t n_pad, RdfParams p,
                                                         int32_t *__restrict__ wrapped,
                                                         float *__restrict__ frame_lo) {
  __shared__ float s_min[3][4];
  __shared__ float s_max[3][4];
  const int f = blockIdx.x;
  const float *X = packed + static_cast<int64_t>(f) * 3 * n_pad;
  const float inf = __builtin_inff();
  bool ok = true, compact = true;
  for (int c = 0; c < 3; ++c) {
    float mn = inf, mx = -inf;
    for (int64_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
      const float v = X[c * n_pad + i];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off));
      mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) {
      s_min[c][threadIdx.x >> 6] = mn;
      s_max[c][threadIdx.x >> 6] = mx;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < 3; ++c) {
      float mn = inf, mx = -inf;
      for (int w = 0; w < 4; ++w) {
        mn = fminf(mn, s_min[c][w]);
        mx = fmaxf(mx, s_max[c][w]);
      }
      // no atom at all: mx - mn = -inf - inf = -inf, counts as compact.  The rounded
      // difference of the extremes bounds every rounded difference (rounding is monotone).
      const float extent = mx - mn;
      ok = ok && (extent < p.t_far[c]) && (p.t_far[c] > 0.0f);
      compact = compact && (extent <= p.box[c]);
      if (frame_lo != nullptr) frame_lo[3 * f + c] = mn;  // origin of the sort's cell grid
    }
    wrapped[f] = ok ? (compact ? 2 : 1) : 0;
  }
}

// Table-driven pair loop over one j-tile, four pairs per iteration.  Equivalent to
// pair_literal for every input; see the file header for the argument.
//   thr[k], k = 0..nbins: smallest float32 squared distance that falls into a bin
//   >= k (thr[0] = 0, thr[nbins] = +inf); p.s_cut: smallest squared distance that
//   fails d < cutoff.
// The j-atoms are the same for every lane, so they are fetched with wave-uniform
// (scalar) loads straight from the packed image, four atoms per component and
// request, one iteration ahead of their use; LDS is left to the histogram.
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool kDiag, bool kWrapped>
__device__ __forceinline__ void table_tile_loop(float xi, float yi, float zi,
                                                const float *__restrict__ xj,
                                                const float *__restrict__ yj,
                                                const float *__restrict__ zj,
                                                uint32_t *__restrict__ hist,
                                                const float *__restrict__ thr,
                                                const RdfParams &p, int tid) {
  // In a diagonal tile lane t only pairs with j > t: a wave skips the j-atoms
  // below its first lane.
  const int j_begin = kDiag ? __builtin_amdgcn_readfirstlane(tid & ~63) : 0;
  float4 nx = *reinterpret_cast<const float4 *>(xj + j_begin);
  float4 ny = *reinterpret_cast<const float4 *>(yj + j_begin);
  float4 nz = *reinterpret_cast<const float4 *>(zj + j_begin);
  for (int j0 = j_begin; j0 < kTile; j0 += 4) {
    const float4 cx = nx, cy = ny, cz = nz;
    // prefetch; the last iteration re-reads its own atoms instead of running past
    // the tile
    const int jn = (j0 + 4 < kTile) ? j0 + 4 : j0;
    nx = *reinterpret_cast<const float4 *>(xj + jn);
    ny = *reinterpret_cast<const float4 *>(yj + jn);
    nz = *reinterpret_cast<const float4 *>(zj + jn);
    const float pxs[4] = {cx.x, cx.y, cx.z, cx.w};
    const float pys[4] = {cy.x, cy.y, cy.z, cy.w};
    const float pzs[4] = {cz.x, cz.y, cz.z, cz.w};
    float s[4];
    int b[4];
    bool keep[4], near_edge[4];
    if (kWrapped) {
      // Two j-atoms per packed instruction: the differences, squares, sums and the
      // bin-position fma each run as one v_pk_*_f32 over an (even, odd) j pair,
      // with the same per-element rounding as the scalar form.
      const v2f xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
      const v2f inv_step2 = {p.inv_step, p.inv_step};
      const v2f eps2 = {p.edge_eps, p.edge_eps};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const v2f rx = v2f{pxs[2 * h], pxs[2 * h + 1]} - xi2;
        const v2f ry = v2f{pys[2 * h], pys[2 * h + 1]} - yi2;
        const v2f rz = v2f{pzs[2 * h], pzs[2 * h + 1]} - zi2;
        // |r| >= t_wrap = nextafter(L/2) selects L - |r| (same magnitude as the
        // literal |r| - L); below it L - |r| >= |r|, so the minimum is exact.
        v2f ax, ay, az;
        ax.x = fminf(fabsf(rx.x), p.box[0] - fabsf(rx.x));
        ax.y = fminf(fabsf(rx.y), p.box[0] - fabsf(rx.y));
        ay.x = fminf(fabsf(ry.x), p.box[1] - fabsf(ry.x));
        ay.y = fminf(fabsf(ry.y), p.box[1] - fabsf(ry.y));
        az.x = fminf(fabsf(rz.x), p.box[2] - fabsf(rz.x));
        az.y = fminf(fabsf(rz.y), p.box[2] - fabsf(rz.y));
        const v2f s2 = (ax * ax + ay * ay) + az * az;
        const v2f root = {__builtin_amdgcn_sqrtf(s2.x), __builtin_amdgcn_sqrtf(s2.y)};
        const v2f t2 = __builtin_elementwise_fma(root, inv_step2, eps2);
        s[2 * h] = s2.x;
        s[2 * h + 1] = s2.y;
        b[2 * h] = static_cast<int>(t2.x);
        b[2 * h + 1] = static_cast<int>(t2.y);
        near_edge[2 * h] = __builtin_amdgcn_fractf(t2.x) < p.edge_eps2;
        near_edge[2 * h + 1] = __builtin_amdgcn_fractf(t2.y) < p.edge_eps2;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        keep[u] = s[u] < p.s_cut;
        if (kDiag) keep[u] = keep[u] && (j0 + u > tid);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float ax = pxs[u] - xi;
        float ay = pys[u] - yi;
        float az = pzs[u] - zi;
        ax = ax - exact_rint_quotient(ax, p.box[0], p.inv_box[0]) * p.box[0];
        ay = ay - exact_rint_quotient(ay, p.box[1], p.inv_box[1]) * p.box[1];
        az = az - exact_rint_quotient(az, p.box[2], p.inv_box[2]) * p.box[2];
        s[u] = (ax * ax + ay * ay) + az * az;
        keep[u] = s[u] < p.s_cut;
        if (kDiag) keep[u] = keep[u] && (j0 + u > tid);
        // bin position shifted up by the error bound eps: away from an edge its
        // integer part is the bin, and "within eps of an edge" is fract < 2 eps
        const float t = __builtin_fmaf(__builtin_amdgcn_sqrtf(s[u]), p.inv_step, p.edge_eps);
        b[u] = static_cast<int>(t);
        near_edge[u] = __builtin_amdgcn_fractf(t) < p.edge_eps2;
      }
    }
    const bool fix = (keep[0] && near_edge[0]) || (keep[1] && near_edge[1]) ||
                     (keep[2] && near_edge[2]) || (keep[3] && near_edge[3]);
    if (__builtin_expect(fix, 0)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (keep[u] && near_edge[u]) {
          int e = min(max(b[u], 0), p.nbins - 1);
          const float lo = thr[e];
          const float hi = thr[e + 1];
          e += (s[u] >= hi) ? 1 : 0;
          e -= (s[u] < lo) ? 1 : 0;
          b[u] = e;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (keep[u]) atomicAdd(&hist[b[u]], 1u);
    }
  }
}


// ----------------------------------------------------------------- fast kernel

// Bits of the float that holds the approximate bin position of one pair, for frames
// whose coordinate differences stay below t_far, about 1.5 box lengths (classes 1 and 2).
//   r = x_j - x_i (the literal difference), then per component
//   m = h - ||r| - h|,  h = box / 2      (= min(|r|, box - |r|) up to two roundings)
//   s = fma(mz, mz, fma(my, my, mx * mx)),  root = v_sqrt_f32(s)       (1 ulp)
//   result = bits(fma(root, nbins / cutoff, 2^k + shift))
// Adding 2^k moves the position's integer part into mantissa bits [F, 23) and its
// fraction, rounded to 2^-F, into bits [0, F).  rdf_fast_setup bounds the distance
// of that position from the literal one by `shift`.  The same function, with the same
// operations, is evaluated by the hot loop (two pairs per packed instruction) and by
// the correction pass; IEEE fma / multiply and the deterministic v_sqrt_f32 give both
// the same bits.
__device__ __forceinline__ uint32_t fast_position_bits(float xi, float yi, float zi, float xj,
                                                       float yj, float zj, float hx, float hy,
                                                       float hz, float scale, float magic,
                                                       int fold) {
  const float rx = xj - xi, ry = yj - yi, rz = zj - zi;
  // bit c of `fold` clear: the hot loop took component c as it is (fold_mask)
  const float mx = (fold & 1) ? hx - fabsf(fabsf(rx) - hx) : rx;
  const float my = (fold & 2) ? hy - fabsf(fabsf(ry) - hy) : ry;
  const float mz = (fold & 4) ? hz - fabsf(fabsf(rz) - hz) : rz;
  const float s = __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
  return __float_as_uint(__builtin_fmaf(__builtin_amdgcn_sqrtf(s), scale, magic));
}

// Literal squared distance of a pair in a wrapped frame (see table_tile_loop).
__device__ __forceinline__ float literal_wrapped_s(float xi, float yi, float zi, float xj,
                                                   float yj, float zj, const float box[3]) {
  const float rx = xj - xi, ry = yj - yi, rz = zj - zi;
  const float ax = fminf(fabsf(rx), box[0] - fabsf(rx));
  const float ay = fminf(fabsf(ry), box[1] - fabsf(ry));
  const float az = fminf(fabsf(rz), box[2] - fabsf(rz));
  return (ax * ax + ay * ay) + az * az;
}

// A value the compiler must keep in a vector register: the simple float32 / integer
// vector instructions issue at half their rate when one operand is a scalar register.
__device__ __forceinline__ float in_vgpr(float v) {
  float out;
  asm volatile("v_mov_b32 %0, %1" : "=v"(out) : "s"(v));
  return out;
}
__device__ __forceinline__ uint32_t in_vgpr(uint32_t v) {
  uint32_t out;
  asm volatile("v_mov_b32 %0, %1" : "=v"(out) : "s"(v));
  return out;
}

// h - ||r| - h| with the absolute values as operand modifiers: two plain v_sub_f32 with
// register operands (the compiler's own choice packs the subtractions and pays for the
// absolute values with separate v_and_b32).
__device__ __forceinline__ float fold_half(float r, float h) {
  float t, m;
  asm("v_sub_f32_e64 %0, |%1|, %2" : "=v"(t) : "v"(r), "v"(h));
  asm("v_sub_f32_e64 %0, %2, |%1|" : "=v"(m) : "v"(t), "v"(h));
  return m;
}

// hist[byte_offset / 4] += value for a histogram that starts at LDS address 0 (the
// kernel has no static LDS, so its dynamic array does; mdx_rdf_create checks that).
// Spares the v_add_u32 of the array's base address per pair.
__device__ __forceinline__ void lds_count(uint32_t byte_offset, uint32_t value) {
  asm volatile("ds_add_u32 %0, %1" : : "v"(byte_offset), "v"(value) : "memory");
}

// Wave-uniform values that the fast kernel keeps in vector registers: the scalar file is
// the scarce one there (eight waves per SIMD), and what the resolving pass reads is used
// lane by lane anyway.
struct FastRegs {
  float box[3];
  float s_cut;
  uint32_t keep_bits;
  uint32_t frac_bits;
  int last_bin;     // nbins - 1
  uint32_t shift;   // frac_bits - 2: position bits -> byte offset of the bin's LDS word
  uint32_t dump;    // the lane's dump position (nbins + 1 + lane, every fraction bit set)
#ifdef MDX_TIMING_SWITCHES
  int drop_noted;   // timing only: noted pairs are never resolved (wrong counts)
#endif
};

// Entries of a wave's queue of noted pairs (16 bits: i-lane << 8 | j-atom).  One iteration
// of the j loop can add 256; full batches of 64 are resolved as soon as there are any, so
// fewer than 64 entries are left when the next iteration starts.
constexpr int kWaveQueue = 256 + 64;

// Resolves n <= 64 noted pairs of a wave, entries [first, first + n) of its queue, one per
// lane: the hot loop has already counted them in the bin of the approximate position (if
// that was below the cutoff); here the literal squared distance decides, and the count is
// moved when the two disagree.  The i-atom of an entry comes from its lane's registers.
__device__ __forceinline__ void fast_resolve_batch(
    int first, int n, const uint16_t *__restrict__ queue, uint32_t masks, int lane, float xi,
    float yi, float zi, const float *__restrict__ xj, const float *__restrict__ yj,
    const float *__restrict__ zj, float hx, float hy, float hz, float scale, float magic,
    uint32_t *__restrict__ hist, const float *__restrict__ g_thr, const FastRegs &r) {
  const int entry = queue[first + (lane < n ? lane : 0)];
  const int j = entry & 255;
  const int i_lane = entry >> 8;
  const float pxi = __shfl(xi, i_lane), pyi = __shfl(yi, i_lane), pzi = __shfl(zi, i_lane);
  if (lane < n) {
    const float pxj = xj[j], pyj = yj[j], pzj = zj[j];
    // the components the hot loop folded for this j-atom's quarter of the tile
    const int fold = static_cast<int>((masks >> (3 * (j >> 6))) & 7u);
    const uint32_t bits =
        fast_position_bits(pxi, pyi, pzi, pxj, pyj, pzj, hx, hy, hz, scale, magic, fold);
    const bool counted = bits < r.keep_bits;
    const int guess = static_cast<int>((bits & 0x7FFFFFu) >> r.frac_bits);
    const float s = literal_wrapped_s(pxi, pyi, pzi, pxj, pyj, pzj, r.box);
    int e = min(max(guess, 0), r.last_bin);
    const float lo = g_thr[e];
    const float hi = g_thr[e + 1];
    e += (s >= hi) ? 1 : 0;
    e -= (s < lo) ? 1 : 0;
    const bool kept = s < r.s_cut;
    if (counted && !(kept && e == guess)) atomicSub(&hist[guess], 1u);
    if (kept && !(counted && e == guess)) atomicAdd(&hist[e], 1u);
  }
}

// Resolves everything the wave has noted (count entries; the end of a tile).
__device__ __forceinline__ void fast_resolve_queue(
    int &count, const uint16_t *__restrict__ queue, uint32_t masks, int lane, float xi, float yi,
    float zi, const float *__restrict__ xj, const float *__restrict__ yj,
    const float *__restrict__ zj, float hx, float hy, float hz, float scale, float magic,
    uint32_t *__restrict__ hist, const float *__restrict__ g_thr, const FastRegs &r) {
#ifdef MDX_TIMING_SWITCHES
  if (r.drop_noted) count = 0;
#endif
  while (count > 0) {
    const int n = min(count, 64);
    count -= n;
    fast_resolve_batch(count, n, queue, masks, lane, xi, yi, zi, xj, yj, zj, hx, hy, hz, scale,
                       magic, hist, g_thr, r);
  }
}

// The j-atoms [j_begin, j_end) of one (tile pair, frame) item: lane = i-atom.  Near-edge
// pairs are noted in the wave's queue (`count` entries so far, uniform over the wave); full
// batches of 64 are resolved on the spot, the rest is left for the caller.
// The loop reads one iteration ahead: on entry nx, ny, nz hold the atoms j_begin .. j_begin
// + 3, on exit the atoms j_end .. j_end + 3 when j_end < j_limit (the caller's next range
// then starts without waiting for a load).  kFold: the components this range folds; masks:
// the same for every quarter of the tile (three bits each), which the resolving pass needs
// to repeat the loop's arithmetic for a noted pair.
template <bool kDiag, int kFold = 7>
__device__ __forceinline__ void fast_j_range(
    int j_begin, int j_end, int j_limit, float4 &nx, float4 &ny, float4 &nz, int &count,
    uint32_t masks, float xi, float yi, float zi,
    const float *__restrict__ xj, const float *__restrict__ yj, const float *__restrict__ zj,
    uint32_t *__restrict__ hist, uint16_t *__restrict__ queue, const float *__restrict__ g_thr,
    const RdfParams &p, const FastRegs &r, int tid, float hx, float hy, float hz, float scale,
    float magic,
    uint32_t addr_mask, uint32_t frac_mask, uint32_t one) {
  const uint32_t shift = r.shift;
  const uint32_t edge = p.fast_edge_units;
  const uint32_t dump = r.dump;
  const v2f xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
  const v2f scale2 = {scale, scale}, magic2 = {magic, magic};
  for (int j0 = j_begin; j0 < j_end; j0 += 4) {
    const float4 cx = nx, cy = ny, cz = nz;
    const int jn = (j0 + 4 < j_limit) ? j0 + 4 : j0;
    nx = *reinterpret_cast<const float4 *>(xj + jn);
    ny = *reinterpret_cast<const float4 *>(yj + jn);
    nz = *reinterpret_cast<const float4 *>(zj + jn);
    const float pxs[4] = {cx.x, cx.y, cx.z, cx.w};
    const float pys[4] = {cy.x, cy.y, cy.z, cy.w};
    const float pzs[4] = {cz.x, cz.y, cz.z, cz.w};
    uint32_t bits[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f rx = v2f{pxs[2 * h], pxs[2 * h + 1]} - xi2;
      const v2f ry = v2f{pys[2 * h], pys[2 * h + 1]} - yi2;
      const v2f rz = v2f{pzs[2 * h], pzs[2 * h + 1]} - zi2;
      v2f mx, my, mz;
      if (kFold & 1) {
        mx.x = fold_half(rx.x, hx);
        mx.y = fold_half(rx.y, hx);
      } else {
        mx = rx;
      }
      if (kFold & 2) {
        my.x = fold_half(ry.x, hy);
        my.y = fold_half(ry.y, hy);
      } else {
        my = ry;
      }
      if (kFold & 4) {
        mz.x = fold_half(rz.x, hz);
        mz.y = fold_half(rz.y, hz);
      } else {
        mz = rz;
      }
      const v2f s2 = __builtin_elementwise_fma(mz, mz, __builtin_elementwise_fma(my, my, mx * mx));
      const v2f root = {__builtin_amdgcn_sqrtf(s2.x), __builtin_amdgcn_sqrtf(s2.y)};
      const v2f t2 = __builtin_elementwise_fma(root, scale2, magic2);
      bits[2 * h] = __float_as_uint(t2.x);
      bits[2 * h + 1] = __float_as_uint(t2.y);
    }
    // A pair beyond the cutoff goes to the lane's dump word (every fraction bit set: never
    // "near an edge"); a NaN (padding) compares above every position and goes there too.
    bool near[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bits[u] = min(bits[u], dump);
      if (kDiag) bits[u] = (j0 + u > tid) ? bits[u] : dump;
      near[u] = (bits[u] & frac_mask) < edge;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) lds_count((bits[u] >> shift) & addr_mask, one);
    // Noting: the wave appends its near-edge pairs to its queue, pair slot by pair slot, the
    // lanes of a slot in lane order (count is uniform over the wave).
    unsigned long long noted[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) noted[u] = __ballot(near[u]);
    if (__builtin_expect((noted[0] | noted[1] | noted[2] | noted[3]) != 0ull, 0)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (noted[u] != 0ull) {
          const uint32_t below = __builtin_amdgcn_mbcnt_hi(
              static_cast<uint32_t>(noted[u] >> 32),
              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(noted[u]), 0u));
          if (near[u])
            queue[count + static_cast<int>(below)] =
                static_cast<uint16_t>(((tid & 63) << 8) | (j0 + u));
          count += __popcll(noted[u]);
        }
      }
#ifdef MDX_TIMING_SWITCHES
      if (r.drop_noted) count = 0;
#endif
      // full batches are resolved at once: fewer than 64 entries stay behind
      while (count >= 64) {
        count -= 64;
        fast_resolve_batch(count, 64, queue, masks, tid & 63, xi, yi, zi, xj, yj, zj, hx, hy, hz,
                           scale, magic, hist, g_thr, r);
      }
    }
  }
}

// One (tile pair, frame) item: all j-atoms of the tile.
template <bool kDiag>
__device__ __forceinline__ void fast_tile_loop(
    float xi, float yi, float zi, const float *__restrict__ xj, const float *__restrict__ yj,
    const float *__restrict__ zj, int j_count, uint32_t *__restrict__ hist,
    uint16_t *__restrict__ queue, const float *__restrict__ g_thr, const RdfParams &p,
    const FastRegs &r, int tid,
    float hx, float hy, float hz, float scale, float magic, uint32_t addr_mask,
    uint32_t frac_mask, uint32_t one) {
  const int j_begin = kDiag ? __builtin_amdgcn_readfirstlane(tid & ~63) : 0;
  const int j_end = (j_count + 3) & ~3;  // padding slots hold NaN and are never counted
  if (j_begin >= j_end) return;
  int count = 0;
  float4 nx = *reinterpret_cast<const float4 *>(xj + j_begin);
  float4 ny = *reinterpret_cast<const float4 *>(yj + j_begin);
  float4 nz = *reinterpret_cast<const float4 *>(zj + j_begin);
  const uint32_t masks = 0xFFFu;  // every component folded
  fast_j_range<kDiag>(j_begin, j_end, j_end, nx, ny, nz, count, masks, xi, yi, zi, xj, yj, zj,
                      hist, queue, g_thr, p, r, tid, hx, hy, hz, scale, magic, addr_mask,
                      frac_mask, one);
  fast_resolve_queue(count, queue, masks, tid & 63, xi, yi, zi, xj, yj, zj, hx, hy, hz, scale,
                     magic, hist, g_thr, r);
}

// Which components of the pairs (64 i-atoms in ibox, 64 j-atoms of a quarter of the j-tile)
// need the minimum-image fold, for the four quarters at once: bits [3q, 3q + 3) of the
// result belong to quarter q, and bit c of them is clear when every difference x_j - x_i of
// component c lies within [-L_c/2, L_c/2].  The boxes are the exact extremes of the float32
// coordinates and float32 subtraction is monotone, so fl(x_j - x_i) <= fl(jhi - ilo) <=
// L_c/2 (and likewise the other way round): the literal rule then rounds r / L_c to zero
// and keeps r itself, and so does the loop variant without the fold -- for such a
// component it is not an approximation at all.  An empty box is (+inf, -inf); its
// differences are -inf (or NaN, which compares false and asks for the fold), and its atoms
// are NaN padding either way.  The 24 conditions (quarter, side, component) are evaluated
// by the first 24 lanes of the wave, one each, and collected with a ballot.
__device__ __forceinline__ uint32_t fold_masks(const float *__restrict__ ibox,
                                               const float *__restrict__ jboxes, int lane,
                                               float hx, float hy, float hz) {
  const int l = lane < 24 ? lane : 0;
  const int q = l / 6, r = l - 6 * q;
  const int side = r >= 3 ? 1 : 0, c = r - 3 * side;
  const float *jbox = jboxes + 6 * q;
  // side 0: highest x_j - lowest x_i; side 1: highest x_i - lowest x_j
  const float high = side ? ibox[3 + c] : jbox[3 + c];
  const float low = side ? jbox[c] : ibox[c];
  const float h = c == 0 ? hx : c == 1 ? hy : hz;
  const unsigned long long ok = __ballot(lane < 24 && high - low <= h);
  const uint32_t bits = static_cast<uint32_t>(ok);
  const uint32_t inside = bits & (bits >> 3);  // bits 6q + c: both sides hold
  uint32_t masks = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) masks |= (~(inside >> (6 * k)) & 7u) << (3 * k);
  return masks;
}

// One (tile pair, frame) item of a sorted frame: the wave's 64 i-atoms and each quarter of
// the j-tile have bounding boxes of their own (ibox, jboxes[4]).  Per quarter the wave
// picks the loop variant that folds only the components that can need it, and, with a
// short cutoff (p.cull_on), skips a quarter out of its reach.  All branches are uniform
// over the wave.  In a diagonal tile pair a wave starts at its own quarter, where only
// j > i counts.
template <bool kDiag>
__device__ __forceinline__ void fast_tile_loop_sorted(
    float xi, float yi, float zi, const float *__restrict__ xj, const float *__restrict__ yj,
    const float *__restrict__ zj, int j_count, uint32_t *__restrict__ hist,
    uint16_t *__restrict__ queue, const float *__restrict__ g_thr, const RdfParams &p,
    const FastRegs &r, int tid,
    float hx, float hy, float hz, float scale, float magic, uint32_t addr_mask,
    uint32_t frac_mask, uint32_t one, const float *__restrict__ ibox,
    const float *__restrict__ jboxes) {
  const int j_end = (j_count + 3) & ~3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int count = 0;
  float4 nx, ny, nz;
  int ahead = -1;  // the j-atom that nx, ny, nz start at
  uint32_t masks = fold_masks(ibox, jboxes, tid & 63, hx, hy, hz);
  // the wave's own quarter of a diagonal tile has two variants only: all or nothing
  if (kDiag && ((masks >> (3 * wave)) & 7u) != 0u) masks |= 7u << (3 * wave);
  for (int q = kDiag ? wave : 0; q < kTile / 64; ++q) {
    const int lo = 64 * q;
    if (lo >= j_end) break;
    const int hi = min(j_end, lo + 64);
    const float *jbox = jboxes + 6 * q;
    if (!kDiag && p.cull_on && box_gap2(ibox, jbox, p) > p.cull_gap2) continue;
#ifdef MDX_TIMING_SWITCHES
    const int mask = (p.timing_mask >= 0 && p.timing_mask < 8)
                         ? p.timing_mask
                         : static_cast<int>((masks >> (3 * q)) & 7u);
#else
    const int mask = static_cast<int>((masks >> (3 * q)) & 7u);
#endif
    if (ahead != lo) {
      nx = *reinterpret_cast<const float4 *>(xj + lo);
      ny = *reinterpret_cast<const float4 *>(yj + lo);
      nz = *reinterpret_cast<const float4 *>(zj + lo);
    }
    ahead = hi;
    if (kDiag && q == wave) {
      // the wave's own quarter: one box on both sides, as a rule narrower than half the box
      if (mask == 0)
        fast_j_range<true, 0>(lo, hi, j_end, nx, ny, nz, count, masks, xi, yi, zi, xj, yj, zj,
                              hist, queue, g_thr, p, r, tid, hx, hy, hz, scale, magic, addr_mask,
                              frac_mask, one);
      else
        fast_j_range<true, 7>(lo, hi, j_end, nx, ny, nz, count, masks, xi, yi, zi, xj, yj, zj,
                              hist, queue, g_thr, p, r, tid, hx, hy, hz, scale, magic, addr_mask,
                              frac_mask, one);
      continue;
    }
#define MDX_FOLD_CASE(M)                                                                     \
  case M:                                                                                    \
    fast_j_range<false, M>(lo, hi, j_end, nx, ny, nz, count, masks, xi, yi, zi, xj, yj, zj, \
                           hist, queue, g_thr, p, r, tid, hx, hy, hz, scale, magic,         \
                           addr_mask, frac_mask, one);                                      \
    break;
    switch (mask) {
      MDX_FOLD_CASE(0)
      MDX_FOLD_CASE(1)
      MDX_FOLD_CASE(2)
      MDX_FOLD_CASE(3)
      MDX_FOLD_CASE(4)
      MDX_FOLD_CASE(5)
      MDX_FOLD_CASE(6)
      MDX_FOLD_CASE(7)
    }
#undef MDX_FOLD_CASE
  }
  fast_resolve_queue(count, queue, masks, tid & 63, xi, yi, zi, xj, yj, zj, hx, hy, hz, scale,
                     magic, hist, g_thr, r);
}

// Persistent blocks with shared work counters.  A chunk is up to frames_per_chunk
// consecutive frames (a frame group) of one tile pair.  The frame groups are dealt to
// eight lists by group % 8, one list per XCD label (blockIdx % 8 tells which blocks share
// an XCD and with it an L2): a block first drains its own label's list, so a frame's
// positions are pulled into one L2 only, then helps the other labels out.  Within a list
// chunks are numbered key by key (the tile pairs are sorted by key), within a key frame
// group by frame group, within a group pair by pair: the blocks of a label work on the
// same few frames at any time (their j-tiles come from that L2), and since they move
// through the keys in step a block flushes its LDS histogram only a few times per launch
// (on a key change and at the end).  Every block claims chunks with one global atomic
// each until all eight lists are empty, so the launch drains evenly whatever the blocks'
// pace.  Frames of class 0 are skipped (the table kernel takes them).
// LDS layout: uint32 hist[lds_hist_words] (bins, the overflow bin, kDumpWords dump words) |
// uint16 queue[4 waves][kWaveQueue] | uint32 claim.
constexpr int kWorkLists = 8;

// The loop needs every wave a SIMD can hold to hide its dependent chains (sub, sub, mul,
// fma, fma, sqrt, fma, shift, and): measured 40.9 k frames/s at four waves per SIMD, 45.5 k at
// five, 48.7 k at seven, 51.9 k at eight.  Left alone the compiler takes 100 scalar
// registers, which caps a SIMD at seven waves (800 scalar registers, allocated in
// sixteens); asked for eight it keeps 72 and moves the rest, set-up values of the outer
// loops, into lanes of a vector register.
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(8, 8))) void rdf_fast_kernel(
    const float *__restrict__ packed, int64_t n_pad, int n_frames, int frames_per_chunk,
    int n_groups, int n_pairs, const TilePair *__restrict__ pairs,
    unsigned int *__restrict__ work_counters, RdfParams p, const float *__restrict__ g_thr,
    const int32_t *__restrict__ frame_wrapped, int lds_hist_words,
    unsigned long long *__restrict__ g_hist, const float *__restrict__ boxes, int n_tiles) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw);
  const int tid = threadIdx.x;
  uint16_t *queue = reinterpret_cast<uint16_t *>(lds_raw + 4 * static_cast<size_t>(lds_hist_words)) +
                    (tid >> 6) * kWaveQueue;
  uint32_t *claim = reinterpret_cast<uint32_t *>(lds_raw + 4 * static_cast<size_t>(lds_hist_words) +
                                                 static_cast<size_t>(kTile / 64) * kWaveQueue * 2);

  const float hx = in_vgpr(p.half_box[0]);
  const float hy = in_vgpr(p.half_box[1]);
  const float hz = in_vgpr(p.half_box[2]);
  const float scale = p.inv_step;
  const float magic = in_vgpr(p.fast_magic);
  const uint32_t addr_mask = in_vgpr(p.fast_addr_mask);
  const uint32_t frac_mask = in_vgpr(p.fast_frac_mask);
  const uint32_t one = in_vgpr(1u);
  FastRegs r;
  for (int c = 0; c < 3; ++c) r.box[c] = in_vgpr(p.box[c]);
  r.s_cut = in_vgpr(p.s_cut);
  r.keep_bits = in_vgpr(p.fast_keep_bits);
  r.frac_bits = in_vgpr(p.fast_frac_bits);
  r.last_bin = static_cast<int>(in_vgpr(static_cast<uint32_t>(p.nbins - 1)));
  r.shift = in_vgpr(p.fast_frac_bits - 2u);
  // position nbins + 1 + lane: 64 distinct words, so the lanes of a wave never collide there
  r.dump = p.fast_dump_bits + (static_cast<uint32_t>(tid & 63) << p.fast_frac_bits);
#ifdef MDX_TIMING_SWITCHES
  r.drop_noted = p.timing_mask == 200;
#endif

  // Adds the block's LDS counts of `key` to the device histogram and clears them.
  auto flush = [&](int key) {
    unsigned long long *out = g_hist + static_cast<int64_t>(key) * p.nbins;
    for (int b = tid; b < lds_hist_words; b += kTile) {
      const uint32_t v = hist[b];
      if (v != 0u && b < p.nbins) atomicAdd(&out[b], static_cast<unsigned long long>(v));
      hist[b] = 0u;
    }
    __syncthreads();
  };

  int cur_key = -1;
  // pairs counted into the LDS bins since they were last cleared: a bin is a uint32, and
  // nothing bounds how many pairs of a block share one bin (coincident atoms, one shell),
  // so the bins are flushed before the count could pass 2^32
  unsigned long long since_flush = 0ull;
  const unsigned long long chunk_pairs_max =
      static_cast<unsigned long long>(frames_per_chunk) * kTile * kTile;
  for (int turn = 0; turn < kWorkLists; ++turn) {
    const int list = (static_cast<int>(blockIdx.x) + turn) % kWorkLists;
    // frame groups list, list + 8, ...: how many, and the list's chunk count
    const uint32_t list_groups =
        n_groups > list ? static_cast<uint32_t>((n_groups - list + kWorkLists - 1) / kWorkLists) : 0u;
    const uint32_t list_chunks = list_groups * static_cast<uint32_t>(n_pairs);
    if (list_chunks == 0u) continue;
    for (;;) {
      if (tid == 0) *claim = atomicAdd(&work_counters[list], 1u);
      __syncthreads();
      const uint32_t chunk = *claim;
      __syncthreads();  // everyone has read the claim before thread 0 writes the next
      if (chunk >= list_chunks) break;
      // chunk / list_groups lies inside the key's range of pairs: that finds the key
      const TilePair in_key = pairs[chunk / list_groups];
      const uint32_t local = chunk - static_cast<uint32_t>(in_key.key_first_pair) * list_groups;
      const uint32_t turn_in_list = local / static_cast<uint32_t>(in_key.key_pairs);
      const int pair_index =
          in_key.key_first_pair +
          static_cast<int>(local - turn_in_list * static_cast<uint32_t>(in_key.key_pairs));
      const int group = list + kWorkLists * static_cast<int>(turn_in_list);
      const TilePair tp = pairs[pair_index];
      if (tp.key != cur_key || since_flush + chunk_pairs_max > 0xFFFFFFFFull) {
        if (cur_key >= 0) {
          flush(cur_key);
        } else {
          for (int b = tid; b < lds_hist_words; b += kTile) hist[b] = 0u;
          __syncthreads();
        }
        cur_key = tp.key;
        since_flush = 0ull;
      }
      since_flush += chunk_pairs_max;
      const int f_begin = group * frames_per_chunk;
      const int f_end = min(n_frames, f_begin + frames_per_chunk);
      const int64_t slot_i = static_cast<int64_t>(tp.tile_i) * kTile + tid;
      for (int f = f_begin; f < f_end; ++f) {
        if (frame_wrapped[f] == 0) continue;  // |r| < t_far only, see rdf_fast_setup
        if (boxes != nullptr && p.cull_on && !tp.diagonal) {
          // short cutoff: the two tiles' boxes are further apart than any pair that counts
          const float *fb = boxes + static_cast<int64_t>(f) * 5 * n_tiles * 6;
          if (box_gap2(fb + tp.tile_i * 6, fb + tp.tile_j * 6, p) > p.cull_gap2) continue;
        }
        const float *X = packed + static_cast<int64_t>(f) * 3 * n_pad;
        const float xi = X[slot_i];
        const float yi = X[n_pad + slot_i];
        const float zi = X[2 * n_pad + slot_i];
        const float *xj = X + static_cast<int64_t>(tp.tile_j) * kTile;
        const float *yj = xj + n_pad;
        const float *zj = yj + n_pad;
        if (boxes != nullptr) {
          const float *quarters = boxes + (static_cast<int64_t>(f) * 5 + 1) * n_tiles * 6;
          const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
          const float *ibox = quarters + (tp.tile_i * 4 + wave) * 6;
          const float *jboxes = quarters + tp.tile_j * 4 * 6;
          if (tp.diagonal)
            fast_tile_loop_sorted<true>(xi, yi, zi, xj, yj, zj, tp.j_count, hist, queue, g_thr, p, r,
                                        tid, hx, hy, hz, scale, magic, addr_mask, frac_mask, one,
                                        ibox, jboxes);
          else
            fast_tile_loop_sorted<false>(xi, yi, zi, xj, yj, zj, tp.j_count, hist, queue, g_thr,
                                         p, r, tid, hx, hy, hz, scale, magic, addr_mask,
                                         frac_mask, one, ibox, jboxes);
        } else if (tp.diagonal)
          fast_tile_loop<true>(xi, yi, zi, xj, yj, zj, tp.j_count, hist, queue, g_thr, p, r, tid,
                               hx, hy, hz, scale, magic, addr_mask, frac_mask, one);
        else
          fast_tile_loop<false>(xi, yi, zi, xj, yj, zj, tp.j_count, hist, queue, g_thr, p, r, tid,
                                hx, hy, hz, scale, magic, addr_mask, frac_mask, one);
      }
      __syncthreads();  // the chunk's counts are in before a flush or the next claim
    }
  }
  if (cur_key >= 0) flush(cur_key);
}

// ---------------------------------------------------------------- tile kernel

// LDS layout: uint32 hist[lds_hist_words] | float thresholds[lds_thr_words] (table
// kernel) | float4 tile[kTile] (literal kernel).
template <bool kLiteral>
__global__ __launch_bounds__(kTile) void rdf_tile_kernel(
    const float *__restrict__ packed, int64_t n_pad, int n_frames, int frames_per_group,
    const TilePair *__restrict__ pairs, RdfParams p,
    const float *__restrict__ g_thresholds, const int32_t *__restrict__ frame_wrapped,
    int lds_hist_words, int lds_thr_words, unsigned long long *__restrict__ g_hist,
    int only_unwrapped) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw);
  float *thr = reinterpret_cast<float *>(lds_raw + 4 * static_cast<size_t>(lds_hist_words));
  float4 *tile = reinterpret_cast<float4 *>(
      lds_raw + 4 * static_cast<size_t>(lds_hist_words + lds_thr_words));

  const int tid = threadIdx.x;
  const TilePair tp = pairs[blockIdx.x];
  const int f_begin = blockIdx.y * frames_per_group;
  const int f_end = min(n_frames, f_begin + frames_per_group);
  if (!kLiteral && only_unwrapped) {
    // complement of the fast kernel: most launches have nothing to do here
    bool any = false;
    for (int f = f_begin; f < f_end; ++f) any = any || (frame_wrapped[f] == 0);
    if (!any) return;
  }

  for (int b = tid; b < lds_hist_words; b += kTile) hist[b] = 0u;
  if (!kLiteral) {
    for (int b = tid; b <= p.nbins; b += kTile) thr[b] = g_thresholds[b];
  }
  __syncthreads();

  const int64_t slot_i = static_cast<int64_t>(tp.tile_i) * kTile + tid;
  const int64_t tile_j0 = static_cast<int64_t>(tp.tile_j) * kTile;

  for (int f = f_begin; f < f_end; ++f) {
    const float *X = packed + static_cast<int64_t>(f) * 3 * n_pad;
    const float xi = X[slot_i];
    const float yi = X[n_pad + slot_i];
    const float zi = X[2 * n_pad + slot_i];
    if (kLiteral) {
      const int64_t slot_j = tile_j0 + tid;
      const float4 mine =
          make_float4(X[slot_j], X[n_pad + slot_j], X[2 * n_pad + slot_j], 0.0f);
      __syncthreads();  // the previous frame's tile reads are complete
      tile[tid] = mine;
      __syncthreads();
      // In a diagonal tile lane t only pairs with j > t: a wave can skip every j
      // that is not above its first lane.
      const int j_start = tp.diagonal ? ((tid & ~63) + 1) : 0;
#pragma unroll 4
      for (int j = j_start; j < kTile; ++j) {
        const bool allowed = !tp.diagonal || (j > tid);
        pair_literal(xi, yi, zi, tile[j], p, allowed, hist);
      }
    } else {
      const float *xj = X + tile_j0;
      const float *yj = xj + n_pad;
      const float *zj = yj + n_pad;
      const bool wrapped = frame_wrapped[f] != 0;
      if (only_unwrapped && frame_wrapped[f] != 0) continue;  // the fast kernel took it
      if (tp.diagonal) {
        if (wrapped)
          table_tile_loop<true, true>(xi, yi, zi, xj, yj, zj, hist, thr, p, tid);
        else
          table_tile_loop<true, false>(xi, yi, zi, xj, yj, zj, hist, thr, p, tid);
      } else {
        if (wrapped)
          table_tile_loop<false, true>(xi, yi, zi, xj, yj, zj, hist, thr, p, tid);
        else
          table_tile_loop<false, false>(xi, yi, zi, xj, yj, zj, hist, thr, p, tid);
      }
    }
  }
  __syncthreads();
  unsigned long long *out = g_hist + static_cast<int64_t>(tp.key) * p.nbins;
  for (int b = tid; b < p.nbins; b += kTile) {
    const uint32_t v = hist[b];
    if (v != 0u) atomicAdd(&out[b], static_cast<unsigned long long>(v));
  }
}

// --------------------------------------------------- host-side threshold table

inline uint32_t float_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
inline float bits_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Literal bin of a non-negative float32 squared distance; nbins means "outside
// the cutoff".  Monotone non-decreasing in s because sqrtf, the float64 divide,
// min and the truncation all are.
inline int literal_bin_of_s(float s, float cutoff, double step, int nbins) {
  volatile float d = sqrtf(s);
  if (!(d < cutoff)) return nbins;
  double q = static_cast<double>(d) / step;
  const double top = static_cast<double>(nbins - 1);
  if (q > top) q = top;
  return static_cast<int>(q);
}

// Smallest non-negative float32 s (by bit pattern order, which is value order for
// non-negative floats) with literal_bin_of_s(s) >= k; +inf bit pattern if none.
inline float first_s_reaching(int k, float cutoff, double step, int nbins) {
  uint32_t lo = 0u;            // candidate range over bit patterns [0, +inf]
  uint32_t hi = 0x7F800000u;   // +inf: sqrtf(inf) = inf, not < cutoff -> nbins >= k
  if (literal_bin_of_s(bits_float(lo), cutoff, step, nbins) >= k) return bits_float(lo);
  while (hi - lo > 1u) {
    const uint32_t mid = lo + (hi - lo) / 2u;
    if (literal_bin_of_s(bits_float(mid), cutoff, step, nbins) >= k)
      hi = mid;
    else
      lo = mid;
  }
  return bits_float(hi);
}

// rint(fl(a / box)) with float32 rounding of the quotient, as the literal kernel.
inline float literal_rint_quotient(float a, float box) {
  volatile float q = a / box;
  return rintf(q);
}

// Smallest non-negative float32 a with literal_rint_quotient(a, box) >= target
// (monotone non-decreasing in a; +inf satisfies it).
inline float first_a_reaching(float target, float box) {
  uint32_t lo = 0u;
  uint32_t hi = 0x7F800000u;
  if (literal_rint_quotient(bits_float(lo), box) >= target) return bits_float(lo);
  while (hi - lo > 1u) {
    const uint32_t mid = lo + (hi - lo) / 2u;
    if (literal_rint_quotient(bits_float(mid), box) >= target)
      hi = mid;
    else
      lo = mid;
  }
  return bits_float(hi);
}

// Rank of every cell of a 2^bits grid along the three-dimensional Hilbert curve
// (Skilling's transpose form: undo the excess work of the Gray code, then interleave the
// bits), for bits = 1..4, laid out as curve_offset() says.  Consecutive ranks are cells
// that share a face; a table that fails this check is replaced by the Z-order, which only
// costs bounding-box quality.
std::vector<uint16_t> hilbert_tables() {
  std::vector<uint16_t> table(kCurveEntries, 0);
  for (int bits = 1; bits <= 4; ++bits) {
    const uint32_t g = 1u << bits;
    uint16_t *out = table.data() + curve_offset(bits);
    std::vector<int32_t> cell_at(static_cast<size_t>(g) * g * g, -1);
    for (uint32_t z = 0; z < g; ++z)
      for (uint32_t y = 0; y < g; ++y)
        for (uint32_t x = 0; x < g; ++x) {
          uint32_t X[3] = {x, y, z};
          const uint32_t M = 1u << (bits - 1);
          for (uint32_t Q = M; Q > 1; Q >>= 1) {
            const uint32_t P = Q - 1;
            for (int i = 0; i < 3; ++i) {
              if (X[i] & Q) {
                X[0] ^= P;
              } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
              }
            }
          }
          for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
          uint32_t t = 0;
          for (uint32_t Q = M; Q > 1; Q >>= 1)
            if (X[2] & Q) t ^= Q - 1;
          for (int i = 0; i < 3; ++i) X[i] ^= t;
          uint32_t rank = 0;
          for (int j = bits - 1; j >= 0; --j)
            for (int i = 0; i < 3; ++i) rank = (rank << 1) | ((X[i] >> j) & 1u);
          const uint32_t cell = (z * g + y) * g + x;
          out[cell] = static_cast<uint16_t>(rank);
          if (rank < cell_at.size()) cell_at[rank] = static_cast<int32_t>(cell);
        }
    bool ok = true;
    for (size_t r = 0; r < cell_at.size() && ok; ++r) {
      if (cell_at[r] < 0) ok = false;
      if (ok && r > 0) {
        const int a = cell_at[r - 1], b = cell_at[r];
        const int dx = std::abs(int(a % g) - int(b % g));
        const int dy = std::abs(int(a / g % g) - int(b / g % g));
        const int dz = std::abs(int(a / (g * g)) - int(b / (g * g)));
        ok = dx + dy + dz == 1;
      }
    }
    if (!ok) {
      for (uint32_t cell = 0; cell < g * g * g; ++cell) {
        const uint32_t x = cell % g, y = cell / g % g, z = cell / (g * g);
        uint32_t code = 0;
        for (int b = 0; b < bits; ++b) {
          code |= ((x >> b) & 1u) << (3 * b);
          code |= ((y >> b) & 1u) << (3 * b + 1);
          code |= ((z >> b) & 1u) << (3 * b + 2);
        }
        out[cell] = static_cast<uint16_t>(code);
      }
    }
  }
  return table;
}

// Parameters of the fast kernel and the bound that makes it exact.
//
// For a pair of a frame whose coordinate differences stay below t_far (|r_c| < 1.5 L_c,
// where the literal rint(r / L) is at most one) let pos_lit = double(d_lit) / step be the
// position the literal sequence bins, and P the number the fast loop forms before adding
// 2^k + shift (fast_position_bits).  With u = 2^-24 and h = L_c / 2 (exact):
//   * per component |m - m_lit| <= u L_c.  h - ||r| - h| equals the magnitude of the
//     minimum image, min(|r|, |L_c - |r||), in exact arithmetic.  Of the fast path's two
//     subtractions at most one rounds: for |r| < h/2, t = |r| - h rounds (by at most
//     u |t| <= u h) and then h/2 < |t| <= h makes m = h - |t| exact; for h/2 <= |r| <= 2h,
//     t is exact (both operands within a factor two) and m rounds by at most u |m| <= u h;
//     for 2h < |r| < 3h, t rounds by at most u |t| <= u L_c and h < |t| <= 2h makes m
//     exact.  The literal value is r itself or r -+ L_c, which is exact too (operands
//     within a factor two) or rounds by at most u h.  Hence the two exact norms differ by
//     at most A = u |L|;
//   * the fast squared sum carries three roundings, (1 + u)^3, so its root 1.5 u, and
//     v_sqrt_f32 is good to one ulp (2 u); the literal squared sum likewise 1.5 u after
//     the root and sqrtf adds u; the scale nbins / cutoff is rounded once (u):
//     |P - pos_lit| <= A / step + 7.1 u pos (7 u and the second-order terms);
//   * the addition rounds to the grid 2^-F: at most 2^-(F+1) more.
// delta is the sum for pos <= nbins + 2 (plus 1e-6 for the top edge, which the literal
// rule places at double(cutoff) / step rather than at nbins).  With shift >= delta the
// stored position lies in [pos_lit, pos_lit + 2 shift]: its integer part is the literal
// bin unless its fraction is below 2 shift, and exactly those pairs are resolved
// literally.
void rdf_fast_setup(RdfParams &p) {
  p.fast_ok = 0;
  for (int c = 0; c < 3; ++c) {
    p.half_box[c] = p.box[c] * 0.5f;
    if (p.half_box[c] * 2.0f != p.box[c]) return;  // not exact (denormal box)
    if (p.t_far[c] <= 0.0f) return;                // no frame may use the wrapped form
  }
  // the mantissa field holds the bins, the overflow bin and the 64 dump words behind it
  int k = 1;
  while ((1 << k) < p.nbins + 2 + kDumpWords) ++k;
  if (k > 15) return;
  const int frac_bits = 23 - k;
  const double u = 0x1p-24;
  const double norm_l = std::sqrt(static_cast<double>(p.box[0]) * p.box[0] +
                                  static_cast<double>(p.box[1]) * p.box[1] +
                                  static_cast<double>(p.box[2]) * p.box[2]);
  const double grid = std::ldexp(1.0, -frac_bits);
  const double delta = u * norm_l / p.step + 7.1 * u * (p.nbins + 2.0) + 0.5 * grid + 1e-6;
  const double units = std::ceil(delta / grid);
  if (2.0 * units > 0.125 / grid) return;  // more than one pair in eight would be resolved
  const uint32_t shift_units = static_cast<uint32_t>(units);
  p.fast_frac_bits = static_cast<uint32_t>(frac_bits);
  p.fast_addr_mask = ((1u << k) - 1u) << 2;
  p.fast_frac_mask = (1u << frac_bits) - 1u;
  p.fast_edge_units = 2u * shift_units;
  p.fast_magic = static_cast<float>(std::ldexp(1.0, k) + shift_units * grid);  // exact
  const float top = static_cast<float>(std::ldexp(1.0, k) + p.nbins);
  p.fast_keep_bits = float_bits(top) + p.fast_edge_units;
  p.fast_dump_bits = float_bits(static_cast<float>(std::ldexp(1.0, k) + p.nbins + 1)) +
                     p.fast_frac_mask;
  p.fast_ok = 1;
}

// Everything of RdfParams that follows from the geometry alone (no device needed).
void rdf_host_params(RdfParams &p, const float box[3], float cutoff, int nbins) {
  for (int c = 0; c < 3; ++c) {
    p.box[c] = box[c];
    p.inv_box[c] = 1.0f / box[c];
  }
  p.cutoff = cutoff;
  p.nbins = nbins;
  // tf.histogram_fixed_width: step = double(range[1] - range[0]) / double(nbins),
  // with the subtraction done in float32.
  const float width = cutoff - 0.0f;
  p.step = static_cast<double>(width) / static_cast<double>(nbins);
  p.inv_step = static_cast<float>(static_cast<double>(nbins) / static_cast<double>(width));
  // |approximate - exact bin position| <= position * (1 ulp sqrt + 2 roundings), kept
  // below position * 2^-21; positions never exceed nbins (+1).
  {
    double eps = (static_cast<double>(nbins) + 2.0) * 0x1p-21 + 0x1p-20;
    if (eps > 0.25) eps = 0.5;  // then every pair consults the thresholds
    p.edge_eps = static_cast<float>(eps);
    p.edge_eps2 = static_cast<float>(2.0 * eps);
  }
  for (int c = 0; c < 3; ++c) {
    p.t_wrap[c] = first_a_reaching(1.0f, box[c]);
    p.t_far[c] = first_a_reaching(2.0f, box[c]);
    // The wrapped-frame path takes min(|r|, L - |r|), which equals the literal
    // result only if no float lies strictly between L/2 and t_wrap (always the case
    // for IEEE float32; checked anyway).  Otherwise no frame may use that path.
    if (p.t_wrap[c] != nextafterf(box[c] * 0.5f, INFINITY)) p.t_far[c] = 0.0f;
  }
  rdf_fast_setup(p);
  p.s_cut = first_s_reaching(nbins, cutoff, p.step, nbins);
}

}  // namespace

struct mdx_rdf_s {
  int device = 0;
  int n_species = 0;
  int n_keys = 0;
  uint32_t flags = 0;
  RdfParams params{};
  std::vector<int64_t> n_per_species;
  std::vector<TilePair> pairs;
  int64_t n_tiles = 0;  // atom tiles
  int64_t n_pad = 0;    // slots
  int lds_hist_words = 0;
  int lds_thr_words = 0;
  size_t lds_bytes = 0;
  DeviceBuffer d_pairs, d_pairs_shard, d_hist, d_thresholds, d_slot_species, d_slot_atom;
  DeviceBuffer d_species_ptr, d_frames, d_packed, d_wrapped, d_counter;
  // frames sorted in space with boxes per tile and per 64 slots: the fast kernel leaves out
  // the fold of components that cannot need it and, for a short cutoff, what is out of reach
  bool sort_frames = false;
  size_t sort_lds_bytes = 0;  // rdf_sort_kernel: five tables of the finest grid's cells
  int n_sort_species = 0;
  DeviceBuffer d_sorted, d_boxes, d_frame_lo, d_sort_species, d_curve;
  int fast_blocks = 0;             // persistent grid of the fast kernel
  size_t fast_lds_bytes = 0;
  int fast_lds_hist_words = 0;
  int shard_rank = 0, shard_count = 1;
  int64_t n_pairs_shard = 0;
  std::vector<const void *> uploaded_ptrs;  // content of d_species_ptr
  void *pinned[2] = {nullptr, nullptr};     // page-locked staging of host input
  size_t pinned_bytes = 0;
  ~mdx_rdf_s() {
    for (void *p : pinned)
      if (p != nullptr) (void)hipHostFree(p);
  }
};

namespace {

int upload_shard(mdx_rdf_s *h, int rank, int count, hipStream_t stream) {
  if (h->d_pairs_shard.ptr != nullptr && h->shard_rank == rank && h->shard_count == count)
    return MDX_OK;
  std::vector<TilePair> mine;
  for (size_t t = 0; t < h->pairs.size(); ++t) {
    if (static_cast<int64_t>(t) % count == rank) mine.push_back(h->pairs[t]);
  }
  for (size_t first = 0; first < mine.size();) {
    size_t last = first;
    while (last < mine.size() && mine[last].key == mine[first].key) ++last;
    for (size_t t = first; t < last; ++t) {
      mine[t].key_first_pair = static_cast<int32_t>(first);
      mine[t].key_pairs = static_cast<int32_t>(last - first);
    }
    first = last;
  }
  MDX_HIP(hipStreamSynchronize(stream));
  MDX_TRY(h->d_pairs_shard.alloc(mine.size() * sizeof(TilePair) + 16));
  if (!mine.empty()) {
    MDX_HIP(hipMemcpy(h->d_pairs_shard.ptr, mine.data(), mine.size() * sizeof(TilePair),
                      hipMemcpyHostToDevice));
  }
  h->shard_rank = rank;
  h->shard_count = count;
  h->n_pairs_shard = static_cast<int64_t>(mine.size());
  return MDX_OK;
}

}  // namespace

namespace {

// Buffers of the spatially sorted copy of `frames` packed frames (no-op without sorting).
int reserve_sorted_frames(mdx_rdf_s *h, int64_t frames) {
  if (!h->sort_frames) return MDX_OK;
  const size_t n = static_cast<size_t>(frames);
  MDX_TRY(h->d_sorted.reserve(sizeof(float) * 3 * static_cast<size_t>(h->n_pad) * n));
  MDX_TRY(h->d_boxes.reserve(sizeof(float) * 6 * 5 * static_cast<size_t>(h->n_tiles) * n + 16));
  MDX_TRY(h->d_frame_lo.reserve(sizeof(float) * 3 * n + 16));
  return MDX_OK;
}

// Packs frames [f0, f0 + nf) of the species arrays behind `species_ptr` (or the listed
// frames) and histograms them with this handle's shard of tile pairs.  d_packed /
// d_wrapped must hold nf frames.  `packed_done`, if given, is recorded once the
// source arrays have been read.
int histogram_frame_block(mdx_rdf_s *h, int dtype, const void *const *species_ptr,
                          const int64_t *d_frames, int64_t f0, int64_t frame_stride, int nf,
                          hipStream_t stream, hipEvent_t packed_done) {
  const bool literal = (h->flags & MDX_RDF_LITERAL_KERNEL) != 0u;
  dim3 pgrid(static_cast<unsigned>(ceil_div(h->n_pad, 64)),
             static_cast<unsigned>(ceil_div(nf, 16)));
  {
    KernelTimer timer(MDX_SLOT_RDF_PACK, stream);
    if (dtype == MDX_F32) {
      hipLaunchKernelGGL(rdf_pack_kernel<float>, pgrid, dim3(256), 0, stream, species_ptr,
                         h->d_slot_species.as<int32_t>(), h->d_slot_atom.as<int32_t>(),
                         d_frames, f0, frame_stride, nf, h->n_pad, h->d_packed.as<float>());
    } else {
      hipLaunchKernelGGL(rdf_pack_kernel<double>, pgrid, dim3(256), 0, stream, species_ptr,
                         h->d_slot_species.as<int32_t>(), h->d_slot_atom.as<int32_t>(),
                         d_frames, f0, frame_stride, nf, h->n_pad, h->d_packed.as<float>());
    }
  }
  MDX_HIP(hipGetLastError());
  if (packed_done != nullptr) MDX_HIP(hipEventRecord(packed_done, stream));
  if (h->n_pairs_shard == 0) return MDX_OK;
  const bool fast = !literal && h->params.fast_ok != 0 &&
                    (h->flags & MDX_RDF_TABLE_KERNEL) == 0u && h->fast_blocks > 0;
  const bool sorted = fast && h->sort_frames && h->n_sort_species > 0;
  if (!literal) {
    hipLaunchKernelGGL(rdf_extent_kernel, dim3(static_cast<unsigned>(nf)), dim3(256), 0,
                       stream, h->d_packed.as<float>(), h->n_pad, h->params,
                       h->d_wrapped.as<int32_t>(), sorted ? h->d_frame_lo.as<float>() : nullptr);
    MDX_HIP(hipGetLastError());
  }
  // the image the pair kernels read: the packed frames, or their spatially sorted copy
  const float *image = h->d_packed.as<float>();
  const float *boxes = nullptr;
  if (sorted) {
    KernelTimer sort_timer(MDX_SLOT_RDF_PACK, stream);
    hipLaunchKernelGGL(rdf_sort_kernel,
                       dim3(static_cast<unsigned>(h->n_sort_species), static_cast<unsigned>(nf)),
                       dim3(256), h->sort_lds_bytes, stream, h->d_packed.as<float>(),
                       h->d_sorted.as<float>(),
                       h->n_pad, static_cast<int>(h->n_tiles), h->d_sort_species.as<SortSpecies>(),
                       h->d_wrapped.as<int32_t>(), h->d_frame_lo.as<float>(),
                       h->d_curve.as<uint16_t>(), h->params, h->d_boxes.as<float>());
    MDX_HIP(hipGetLastError());
    image = h->d_sorted.as<float>();
    boxes = h->d_boxes.as<float>();
  }
  KernelTimer timer(MDX_SLOT_RDF_TILE, stream);
  if (fast) {
    // chunks of about 10 us of work: short against the launch, long against a claim
    int frames_per_chunk = 8;
    if (const char *env = getenv("MDX_RDF_FRAMES_PER_CHUNK")) {
      const int v = atoi(env);
      if (v >= 1) frames_per_chunk = v;
    }
    // a chunk's pairs must fit the kernel's uint32 LDS bins with room to spare
    if (frames_per_chunk > 16384) frames_per_chunk = 16384;
    const int64_t n_groups = ceil_div(nf, frames_per_chunk);
    const int64_t n_chunks = n_groups * h->n_pairs_shard;
    MDX_REQUIRE(n_chunks + h->fast_blocks < (int64_t(1) << 31), "too many work chunks");
    MDX_TRY(h->d_counter.reserve(sizeof(unsigned int) * kWorkLists));
    MDX_HIP(hipMemsetAsync(h->d_counter.ptr, 0, sizeof(unsigned int) * kWorkLists, stream));
    const int64_t blocks = std::min<int64_t>(h->fast_blocks, std::max<int64_t>(n_chunks, kWorkLists));
    hipLaunchKernelGGL(rdf_fast_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kTile),
                       h->fast_lds_bytes, stream, image, h->n_pad, nf,
                       frames_per_chunk, static_cast<int>(n_groups),
                       static_cast<int>(h->n_pairs_shard), h->d_pairs_shard.as<TilePair>(),
                       h->d_counter.as<unsigned int>(), h->params,
                       h->d_thresholds.as<float>(), h->d_wrapped.as<int32_t>(),
                       h->fast_lds_hist_words, h->d_hist.as<unsigned long long>(), boxes,
                       static_cast<int>(h->n_tiles));
    MDX_HIP(hipGetLastError());
  }
  // One block per (tile pair, frame group).  Frames per block: large enough to amortise
  // the histogram flush, small enough that the last round of blocks is short against
  // the launch.  After the fast kernel this launch only takes the frames that one left
  // out (coordinates spread over more than 1.5 box lengths); its blocks return at once
  // when their group has none.
  int64_t fpg = (static_cast<int64_t>(nf) * h->n_pairs_shard) / 65536;
  if (fpg < 1) fpg = 1;
  if (fpg > kMaxFramesPerGroup) fpg = kMaxFramesPerGroup;
  if (fast) fpg = kMaxFramesPerGroup;
  if (const char *env = getenv("MDX_RDF_FRAMES_PER_GROUP")) {
    const int v = atoi(env);
    if (v >= 1 && v <= kMaxFramesPerGroup) fpg = v;
  }
  dim3 grid(static_cast<unsigned>(h->n_pairs_shard),
            static_cast<unsigned>(ceil_div(nf, fpg)));
  if (literal) {
    hipLaunchKernelGGL(rdf_tile_kernel<true>, grid, dim3(kTile), h->lds_bytes, stream,
                       h->d_packed.as<float>(), h->n_pad, nf, static_cast<int>(fpg),
                       h->d_pairs_shard.as<TilePair>(), h->params,
                       static_cast<const float *>(nullptr),
                       static_cast<const int32_t *>(nullptr), h->lds_hist_words,
                       h->lds_thr_words, h->d_hist.as<unsigned long long>(), 0);
  } else {
    hipLaunchKernelGGL(rdf_tile_kernel<false>, grid, dim3(kTile), h->lds_bytes, stream,
                       image, h->n_pad, nf, static_cast<int>(fpg),
                       h->d_pairs_shard.as<TilePair>(), h->params,
                       h->d_thresholds.as<float>(), h->d_wrapped.as<int32_t>(),
                       h->lds_hist_words, h->lds_thr_words,
                       h->d_hist.as<unsigned long long>(), fast ? 1 : 0);
  }
  MDX_HIP(hipGetLastError());
  return MDX_OK;
}

// Events of the copy/compute hand-over; on every exit path both streams are drained
// before the staging blocks go back to the cache.
struct EventPairGuard {
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t streams[2] = {nullptr, nullptr};
  ~EventPairGuard() {
    for (hipStream_t s : streams)
      if (s != nullptr) (void)hipStreamSynchronize(s);
    for (hipEvent_t e : ev)
      if (e != nullptr) (void)hipEventDestroy(e);
  }
};

// dst[a][0..row) = src[a * pitch ..): the block's frames of every atom, gathered into the
// page-locked image.  A block of 10 000 atoms x 256 frames is 30 MB in 10 000 pieces; one
// host thread moves that in about the time the GPU needs for the block, so large
// gathers are split over a few threads.
void gather_rows(char *dst, const char *src, int64_t n_rows, size_t row, size_t pitch,
                 unsigned max_threads = 4) {
  auto part = [=](int64_t a0, int64_t a1) {
    for (int64_t a = a0; a < a1; ++a)
      memcpy(dst + static_cast<size_t>(a) * row, src + static_cast<size_t>(a) * pitch, row);
  };
  const size_t bytes = static_cast<size_t>(n_rows) * row;
  unsigned n_threads = bytes >= (size_t(4) << 20) ? max_threads : 1u;
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw != 0 && hw < n_threads) n_threads = hw;
  if (n_threads <= 1 || n_rows < 64) {
    part(0, n_rows);
    return;
  }
  std::vector<std::thread> workers;
  const int64_t share = ceil_div(n_rows, static_cast<int64_t>(n_threads));
  for (unsigned t = 1; t < n_threads; ++t) {
    const int64_t a0 = std::min<int64_t>(n_rows, share * t);
    const int64_t a1 = std::min<int64_t>(n_rows, share * (t + 1));
    if (a1 > a0) workers.emplace_back(part, a0, a1);
  }
  part(0, std::min<int64_t>(n_rows, share));
  for (std::thread &w : workers) w.join();
}

int accumulate_host_pipelined(mdx_rdf_s *h, const void *const *species_pos, int dtype,
                              int64_t frame_stride, int64_t n_frames, hipStream_t stream) {
  hipStream_t copy_stream;
  MDX_TRY(get_copy_stream(&copy_stream));
  const size_t elem = dtype_size(dtype);
  // block size: a quarter of the call, between 64 and 512 frames
  int64_t block = n_frames / 4;
  if (block < 64) block = 64;
  if (block > 512) block = 512;
  if (const char *env = getenv("MDX_RDF_HOST_BLOCK")) {  // A/B timing of the staging block
    const int v = atoi(env);
    if (v >= 16 && v <= 16384) block = v;
  }
  unsigned gather_threads = 4;
  if (const char *env = getenv("MDX_RDF_GATHER_THREADS")) {
    const int v = atoi(env);
    if (v >= 1 && v <= 64) gather_threads = static_cast<unsigned>(v);
  }
  const size_t bytes_per_frame = sizeof(float) * 3 * static_cast<size_t>(h->n_pad);
  MDX_TRY(h->d_packed.reserve(bytes_per_frame * static_cast<size_t>(block)));
  MDX_TRY(h->d_wrapped.reserve(sizeof(int32_t) * static_cast<size_t>(block)));
  MDX_TRY(reserve_sorted_frames(h, block));

  // One staging image per block parity: all species back to back, each as
  // [atoms][block][3]; a page-locked twin on the host is filled row by row (a row =
  // the block's frames of one atom, contiguous in the caller's array) and sent with
  // one asynchronous copy.
  const int ns = h->n_species;
  std::vector<size_t> offset(ns + 1, 0);
  for (int s = 0; s < ns; ++s) {
    if (h->n_per_species[s] > 0)
      MDX_REQUIRE(species_pos[s] != nullptr, "positions of species %d are null", s);
    offset[s + 1] = offset[s] + static_cast<size_t>(h->n_per_species[s]) * block * 3 * elem;
  }
  const size_t image_bytes = offset[ns];
  if (h->pinned_bytes < image_bytes) {
    MDX_HIP(hipStreamSynchronize(copy_stream));
    for (void *&p : h->pinned) {
      if (p != nullptr) (void)hipHostFree(p);
      p = nullptr;
    }
    h->pinned_bytes = 0;
    for (void *&p : h->pinned) {
      note_driver_allocation();
      MDX_HIP(hipHostMalloc(&p, image_bytes, hipHostMallocDefault));
    }
    h->pinned_bytes = image_bytes;
  }
  DeviceBuffer staged[2];
  DeviceBuffer d_table;
  EventPairGuard events;  // [0,1] copied, [2,3] consumed; declared after the buffers
  events.streams[0] = stream;
  events.streams[1] = copy_stream;
  std::vector<const void *> table(2 * static_cast<size_t>(ns), nullptr);
  for (int b = 0; b < 2; ++b) {
    MDX_TRY(staged[b].alloc(image_bytes));
    for (int s = 0; s < ns; ++s)
      table[b * ns + s] = static_cast<const char *>(staged[b].ptr) + offset[s];
  }
  MDX_TRY(d_table.alloc(sizeof(void *) * table.size()));
  MDX_HIP(hipMemcpyAsync(d_table.ptr, table.data(), sizeof(void *) * table.size(),
                         hipMemcpyHostToDevice, stream));
  for (hipEvent_t &e : events.ev) MDX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // the staging blocks may have been used by earlier work on the compute stream
  MDX_HIP(hipEventRecord(events.ev[2], stream));
  MDX_HIP(hipStreamWaitEvent(copy_stream, events.ev[2], 0));

  int64_t index = 0;
  int64_t step = 0;
  for (int64_t f0 = 0; f0 < n_frames; f0 += step, ++index) {
    const int b = static_cast<int>(index & 1);
    // a short first block: its copy is the only one no kernel hides
    step = (index == 0) ? std::min<int64_t>(block, 48) : block;
    const int nf = static_cast<int>(std::min<int64_t>(step, n_frames - f0));
    if (index >= 2) {
      // the page-locked twin is free once its copy has left, the device image once
      // the pack kernel has read it
      MDX_HIP(hipEventSynchronize(events.ev[b]));
      MDX_HIP(hipStreamWaitEvent(copy_stream, events.ev[2 + b], 0));
    }
    const size_t row = static_cast<size_t>(nf) * 3 * elem;
    char *twin = static_cast<char *>(h->pinned[b]);
    for (int s = 0; s < ns; ++s) {
      const char *src = static_cast<const char *>(species_pos[s]) + f0 * 3 * elem;
      char *dst = twin + offset[s];
      const size_t pitch = static_cast<size_t>(frame_stride) * 3 * elem;
      gather_rows(dst, src, h->n_per_species[s], row, pitch, gather_threads);
    }
    MDX_HIP(hipMemcpyAsync(staged[b].ptr, twin, image_bytes, hipMemcpyHostToDevice,
                           copy_stream));
    MDX_HIP(hipEventRecord(events.ev[b], copy_stream));
    MDX_HIP(hipStreamWaitEvent(stream, events.ev[b], 0));
    MDX_TRY(histogram_frame_block(h, dtype, d_table.as<const void *>() + b * ns, nullptr, 0, nf,
                                  nf, stream, events.ev[2 + b]));
  }
  MDX_HIP(hipStreamSynchronize(stream));
  MDX_HIP(hipStreamSynchronize(copy_stream));
  return MDX_OK;
}

}  // namespace

extern "C" {

int mdx_rdf_create(mdx_rdf_t *handle, int n_species, const int64_t *n_per_species,
                   const float box[3], float cutoff, int nbins, uint32_t flags) {
  MDX_REQUIRE(handle != nullptr, "handle is null");
  *handle = nullptr;
  MDX_REQUIRE(n_species >= 1 && n_species <= 1024, "n_species out of range");
  MDX_REQUIRE(n_per_species != nullptr && box != nullptr, "null argument");
  MDX_REQUIRE(nbins >= 1, "nbins must be positive");
  MDX_REQUIRE(cutoff > 0.0f && std::isfinite(cutoff), "cutoff must be positive and finite");
  for (int c = 0; c < 3; ++c)
    MDX_REQUIRE(box[c] > 0.0f && std::isfinite(box[c]), "box lengths must be positive");
  for (int s = 0; s < n_species; ++s)
    MDX_REQUIRE(n_per_species[s] >= 0 && n_per_species[s] < (int64_t(1) << 31),
                "species size out of range");

  mdx_rdf_s *h = new (std::nothrow) mdx_rdf_s;
  if (!h) return fail(MDX_ERR_OUT_OF_MEMORY, "out of host memory");
  struct Guard {
    mdx_rdf_s *h;
    ~Guard() { delete h; }
  } guard{h};

  MDX_HIP(hipGetDevice(&h->device));
  h->n_species = n_species;
  h->n_keys = n_species * (n_species + 1) / 2;
  h->flags = flags;
  h->n_per_species.assign(n_per_species, n_per_species + n_species);
  RdfParams &p = h->params;
  rdf_host_params(p, box, cutoff, nbins);

  // slots: per species, the atoms that take part, padded to whole tiles
  const bool strict = (flags & MDX_RDF_STRICT_REFERENCE_MASKS) != 0u;
  std::vector<int32_t> slot_species, slot_atom;
  std::vector<int64_t> tile_first(n_species), tile_count(n_species);
  int64_t tiles = 0;
  for (int s = 0; s < n_species; ++s) {
    const int64_t first = strict ? 1 : 0;
    const int64_t used = n_per_species[s] > first ? n_per_species[s] - first : 0;
    const int64_t nt = ceil_div(used, kTile);
    tile_first[s] = tiles;
    tile_count[s] = nt;
    for (int64_t t = 0; t < nt * kTile; ++t) {
      if (t < used) {
        slot_species.push_back(s);
        slot_atom.push_back(static_cast<int32_t>(first + t));
      } else {
        slot_species.push_back(-1);
        slot_atom.push_back(0);
      }
    }
    tiles += nt;
  }
  h->n_tiles = tiles;
  h->n_pad = tiles * kTile;

  // tile pairs in key order (a,b), a <= b
  int key = 0;
  for (int a = 0; a < n_species; ++a) {
    for (int b = a; b < n_species; ++b, ++key) {
      for (int64_t ti = 0; ti < tile_count[a]; ++ti) {
        const int64_t tj0 = (a == b) ? ti : 0;
        for (int64_t tj = tj0; tj < tile_count[b]; ++tj) {
          TilePair tp;
          tp.tile_i = static_cast<int32_t>(tile_first[a] + ti);
          tp.tile_j = static_cast<int32_t>(tile_first[b] + tj);
          tp.key = key;
          tp.diagonal = (a == b && ti == tj) ? 1 : 0;
          {
            const int64_t first = strict ? 1 : 0;
            const int64_t used = n_per_species[b] > first ? n_per_species[b] - first : 0;
            const int64_t left = used - tj * kTile;
            tp.j_count = static_cast<int32_t>(left < kTile ? left : kTile);
          }
          tp.key_first_pair = 0;
          tp.key_pairs = 1;
          tp.pad_ = 0;
          h->pairs.push_back(tp);
        }
      }
    }
  }

  std::vector<SortSpecies> sort_table;
  // Frames sorted in space serve two ends.  (1) Short cutoff against the box: skip what is
  // out of reach (whole tile pairs, and quarters of a j-tile per wave).  The 64 atoms of a
  // wave or of a quarter, of a species with n atoms, fill about the 64/n-th part of the box;
  // skipping can only pay when the cutoff plus the edge of such a group stays well below
  // half the box (measured: 10k-atom melt in a 65 box, cutoff 10: 74 k instead of 57 k
  // frames/s; 100k-atom water box, cutoff 10: 4.4 k instead of 0.6 k).  (2) Any cutoff: a
  // component in which two such groups lie within half a box length of each other needs no
  // minimum-image fold (fold_mask), which pays as soon as the groups are small against the
  // box (10k-atom melt: 56 % of the components).  MDX_RDF_CULL=0/1 and MDX_RDF_UNFOLD=0/1
  // override the choices (the counts do not change).
  {
    const double volume = static_cast<double>(box[0]) * box[1] * box[2];
    const double shortest = std::min({box[0], box[1], box[2]});
    bool worth = false, unfold = false;
    std::vector<SortSpecies> table;
    for (int s = 0; s < n_species; ++s) {
      if (tile_count[s] == 0) continue;
      const int64_t first = strict ? 1 : 0;
      const int64_t used = n_per_species[s] - first;
      SortSpecies sp;
      sp.first_slot = static_cast<int32_t>(tile_first[s] * kTile);
      sp.used = static_cast<int32_t>(used);
      sp.tiles = static_cast<int32_t>(tile_count[s]);
      sp.cell_bits = 1;  // four to thirty-two atoms per cell, 16 cells per axis at most
      while (sp.cell_bits < 4 && (int64_t(1) << (3 * (sp.cell_bits + 1))) * 4 <= used)
        ++sp.cell_bits;
      table.push_back(sp);
      const double group_edge = std::cbrt(volume * std::min<double>(1.0, 64.0 / used));
      if (tile_count[s] >= 8 && cutoff + group_edge < 0.45 * shortest) worth = true;
      if (tile_count[s] >= 8 && group_edge < 0.3 * shortest) unfold = true;
    }
    if (const char *env = getenv("MDX_RDF_CULL")) worth = atoi(env) != 0;
    if (const char *env = getenv("MDX_RDF_UNFOLD")) unfold = atoi(env) != 0;
    h->sort_frames =
        (worth || unfold) && !table.empty() && (flags & MDX_RDF_LITERAL_KERNEL) == 0u;
    p.cull_on = (h->sort_frames && worth) ? 1 : 0;
    if (h->sort_frames) sort_table = table;  // uploaded with the other device tables below
    for (const SortSpecies &sp : sort_table)
      h->sort_lds_bytes = std::max(
          h->sort_lds_bytes,
          sizeof(uint32_t) * 5 * (size_t(1) << (3 * sp.cell_bits)) +
              (sp.used <= kSortIndexed ? sizeof(uint16_t) * static_cast<size_t>(sp.used) : 0) + 16);
    if (h->sort_lds_bytes > 0)
      MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rdf_sort_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(h->sort_lds_bytes)));
    // A skipped tile pair must hold no pair the literal rule would count.  Its pairs are
    // at least the box gap apart; the literal float32 distance is below the true one by
    // a few 2^-24 of itself plus 2^-24 of the box per component at most, and the gap's
    // own arithmetic errs by as little: a margin of 1e-5 of cutoff plus box diagonal is
    // hundreds of times that.
    const double norm_l = std::sqrt(double(box[0]) * box[0] + double(box[1]) * box[1] +
                                    double(box[2]) * box[2]);
    const double reach = static_cast<double>(cutoff) * (1.0 + 1e-5) + 1e-5 * norm_l;
    p.cull_gap2 = static_cast<float>(reach * reach * (1.0 + 1e-6));
  }

  // LDS budget
  const bool literal = (flags & MDX_RDF_LITERAL_KERNEL) != 0u;
  h->lds_hist_words = (nbins + 3) & ~3;
  h->lds_thr_words = literal ? 0 : ((nbins + 1 + 3) & ~3);
  h->lds_bytes = 4 * static_cast<size_t>(h->lds_hist_words + h->lds_thr_words) +
                 (literal ? sizeof(float4) * kTile : 0);
  if (h->lds_bytes > 160 * 1024)
    return fail(MDX_ERR_INVALID_ARGUMENT,
                "nbins=%d needs %zu bytes of LDS per workgroup, more than the 160 KiB of a "
                "gfx950 compute unit",
                nbins, h->lds_bytes);
  if (literal) {
    MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rdf_tile_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(h->lds_bytes)));
  } else {
    MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rdf_tile_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(h->lds_bytes)));
  }

  if (!literal && p.fast_ok) {
    h->fast_lds_hist_words = (nbins + 1 + kDumpWords + 3) & ~3;
    h->fast_lds_bytes = 4 * static_cast<size_t>(h->fast_lds_hist_words) +
                        static_cast<size_t>(kTile / 64) * kWaveQueue * sizeof(uint16_t) + 16;
    const void *fast_kernel = reinterpret_cast<const void *>(rdf_fast_kernel);
    if (h->fast_lds_bytes <= 160 * 1024) {
      MDX_HIP(hipFuncSetAttribute(fast_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(h->fast_lds_bytes)));
      hipFuncAttributes attr;
      MDX_HIP(hipFuncGetAttributes(&attr, fast_kernel));
      int per_cu = 0;
      if (attr.sharedSizeBytes == 0)  // the histogram then starts at LDS address 0
        MDX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &per_cu, fast_kernel, kTile, h->fast_lds_bytes));
      hipDeviceProp_t prop;
      MDX_HIP(hipGetDeviceProperties(&prop, h->device));
      h->fast_blocks = per_cu * prop.multiProcessorCount;
#ifdef MDX_TIMING_SWITCHES
      // builds for timing experiments only: without the near-edge test the counts are wrong
      if (getenv("MDX_RDF_FAST_NO_EDGE")) p.fast_edge_units = 0;
      p.timing_mask = getenv("MDX_RDF_TIMING_MASK") ? atoi(getenv("MDX_RDF_TIMING_MASK")) : -1;
#endif
      if (getenv("MDX_RDF_DEBUG"))
        fprintf(stderr, "rdf fast: %d blocks/CU, %d CUs, lds %zu, edge units %u of %u\n", per_cu,
                prop.multiProcessorCount, h->fast_lds_bytes, p.fast_edge_units,
                1u << p.fast_frac_bits);
      if (const char *env = getenv("MDX_RDF_FAST_BLOCKS")) {
        const int v = atoi(env);
        if (v >= 1) h->fast_blocks = v;
      }
    }
  }

  // device tables
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  // The tables below are filled with blocking copies, which are not ordered against
  // the library's non-blocking stream: a recycled pool block may still be read by
  // work queued there, so let that work finish first.
  MDX_HIP(hipStreamSynchronize(stream));
  MDX_TRY(h->d_hist.alloc(sizeof(uint64_t) * static_cast<size_t>(h->n_keys) * nbins));
  MDX_HIP(hipMemsetAsync(h->d_hist.ptr, 0, h->d_hist.bytes, stream));
  MDX_TRY(h->d_slot_species.alloc(sizeof(int32_t) * slot_species.size() + 16));
  MDX_TRY(h->d_slot_atom.alloc(sizeof(int32_t) * slot_atom.size() + 16));
  if (!slot_species.empty()) {
    MDX_HIP(hipMemcpy(h->d_slot_species.ptr, slot_species.data(),
                      sizeof(int32_t) * slot_species.size(), hipMemcpyHostToDevice));
    MDX_HIP(hipMemcpy(h->d_slot_atom.ptr, slot_atom.data(),
                      sizeof(int32_t) * slot_atom.size(), hipMemcpyHostToDevice));
  }
  MDX_TRY(h->d_species_ptr.alloc(sizeof(void *) * n_species));
  if (!sort_table.empty()) {
    h->n_sort_species = static_cast<int>(sort_table.size());
    MDX_TRY(h->d_sort_species.alloc(sizeof(SortSpecies) * sort_table.size()));
    MDX_HIP(hipMemcpy(h->d_sort_species.ptr, sort_table.data(),
                      sizeof(SortSpecies) * sort_table.size(), hipMemcpyHostToDevice));
    const std::vector<uint16_t> curve = hilbert_tables();
    MDX_TRY(h->d_curve.alloc(sizeof(uint16_t) * curve.size()));
    MDX_HIP(hipMemcpy(h->d_curve.ptr, curve.data(), sizeof(uint16_t) * curve.size(),
                      hipMemcpyHostToDevice));
  }

  // thresholds of the table kernel
  if (!literal) {
    std::vector<float> thr(static_cast<size_t>(nbins) + 1);
    thr[0] = 0.0f;
    for (int k = 1; k < nbins; ++k) thr[k] = first_s_reaching(k, cutoff, p.step, nbins);
    thr[nbins] = bits_float(0x7F800000u);
    MDX_TRY(h->d_thresholds.alloc(sizeof(float) * thr.size()));
    MDX_HIP(hipMemcpy(h->d_thresholds.ptr, thr.data(), sizeof(float) * thr.size(),
                      hipMemcpyHostToDevice));
  }
  MDX_HIP(hipStreamSynchronize(stream));

  guard.h = nullptr;
  *handle = h;
  return MDX_OK;
}

int mdx_rdf_fast_plan(const float box[3], float cutoff, int nbins, mdx_rdf_fast_plan_t *plan) {
  MDX_REQUIRE(box != nullptr && plan != nullptr, "null argument");
  MDX_REQUIRE(nbins >= 1, "nbins must be positive");
  MDX_REQUIRE(cutoff > 0.0f && std::isfinite(cutoff), "cutoff must be positive and finite");
  for (int c = 0; c < 3; ++c)
    MDX_REQUIRE(box[c] > 0.0f && std::isfinite(box[c]), "box lengths must be positive");
  RdfParams p{};
  rdf_host_params(p, box, cutoff, nbins);
  memset(plan, 0, sizeof(*plan));
  plan->usable = p.fast_ok;
  plan->step = p.step;
  plan->scale = p.inv_step;
  plan->s_cut = p.s_cut;
  for (int c = 0; c < 3; ++c) plan->half_box[c] = p.half_box[c];
  if (p.fast_ok) {
    plan->frac_bits = static_cast<int>(p.fast_frac_bits);
    plan->edge_units = p.fast_edge_units;
    plan->keep_bits = p.fast_keep_bits;
    plan->magic = p.fast_magic;
  }
  return MDX_OK;
}

int mdx_rdf_destroy(mdx_rdf_t handle) {
  if (!handle) return MDX_OK;
  hipStream_t stream;
  if (get_stream(&stream) == MDX_OK) (void)hipStreamSynchronize(stream);
  delete handle;
  return MDX_OK;
}

int mdx_rdf_reset(mdx_rdf_t h) {
  MDX_REQUIRE(h != nullptr, "handle is null");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  MDX_HIP(hipMemsetAsync(h->d_hist.ptr, 0, h->d_hist.bytes, stream));
  return MDX_OK;
}

int mdx_rdf_num_keys(mdx_rdf_t h, int *n_keys) {
  MDX_REQUIRE(h != nullptr && n_keys != nullptr, "null argument");
  *n_keys = h->n_keys;
  return MDX_OK;
}

int mdx_rdf_num_tiles(mdx_rdf_t h, int64_t *n_tiles) {
  MDX_REQUIRE(h != nullptr && n_tiles != nullptr, "null argument");
  *n_tiles = static_cast<int64_t>(h->pairs.size());
  return MDX_OK;
}

int mdx_rdf_accumulate(mdx_rdf_t h, const void *const *species_pos, int dtype,
                       int location, int64_t frame_stride, const int64_t *frames,
                       int64_t n_frames, int shard_rank, int shard_count) {
  MDX_REQUIRE(h != nullptr && species_pos != nullptr, "null argument");
  MDX_REQUIRE(dtype == MDX_F32 || dtype == MDX_F64, "dtype must be MDX_F32 or MDX_F64");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  MDX_REQUIRE(n_frames >= 0 && frame_stride >= 0, "negative frame count");
  MDX_REQUIRE(shard_count >= 1 && shard_rank >= 0 && shard_rank < shard_count,
              "bad shard (rank %d of %d)", shard_rank, shard_count);
  if (frames != nullptr) {
    for (int64_t i = 0; i < n_frames; ++i)
      MDX_REQUIRE(frames[i] >= 0 && frames[i] < frame_stride,
                  "frame index %lld outside [0, %lld)", (long long)frames[i],
                  (long long)frame_stride);
  } else {
    MDX_REQUIRE(n_frames <= frame_stride, "n_frames exceeds frame_stride");
  }
  if (n_frames == 0 || h->n_pad == 0) return MDX_OK;

  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  MDX_TRY(upload_shard(h, shard_rank, shard_count, stream));

  // Host input with a contiguous frame range: blocks of frames are copied on a
  // second stream into two alternating staging sets while the previous block is
  // being histogrammed.
  // A range shorter than the arrays always goes this way, whatever its length: the
  // rows are copied frame range by frame range, never past the end of the caller's array.
  if (location == MDX_HOST && frames == nullptr && (n_frames >= 128 || n_frames < frame_stride))
    return accumulate_host_pipelined(h, species_pos, dtype, frame_stride, n_frames, stream);

  // species arrays on the device
  std::vector<DeviceBuffer> staged(location == MDX_HOST ? h->n_species : 0);
  std::vector<const void *> ptrs(h->n_species, nullptr);
  const size_t elem = dtype_size(dtype);
  for (int s = 0; s < h->n_species; ++s) {
    if (h->n_per_species[s] == 0) continue;
    MDX_REQUIRE(species_pos[s] != nullptr, "positions of species %d are null", s);
    const size_t bytes =
        static_cast<size_t>(h->n_per_species[s]) * frame_stride * 3 * elem;
    if (location == MDX_HOST) {
      MDX_TRY(to_device(species_pos[s], MDX_HOST, bytes, staged[s], stream, &ptrs[s]));
    } else {
      ptrs[s] = species_pos[s];
    }
  }
  // the pointer table and frame list may still be in use by an earlier call
  if (ptrs != h->uploaded_ptrs) {
    MDX_HIP(hipStreamSynchronize(stream));
    MDX_HIP(hipMemcpy(h->d_species_ptr.ptr, ptrs.data(), sizeof(void *) * h->n_species,
                      hipMemcpyHostToDevice));
    h->uploaded_ptrs = ptrs;
  }
  const int64_t *d_frames = nullptr;
  if (frames != nullptr) {
    MDX_HIP(hipStreamSynchronize(stream));
    MDX_TRY(h->d_frames.reserve(sizeof(int64_t) * static_cast<size_t>(n_frames)));
    MDX_HIP(hipMemcpy(h->d_frames.ptr, frames, sizeof(int64_t) * n_frames,
                      hipMemcpyHostToDevice));
    d_frames = h->d_frames.as<int64_t>();
  }

  // frames are processed in chunks bounded by the packed image size
  const size_t bytes_per_frame = sizeof(float) * 3 * static_cast<size_t>(h->n_pad);
  int64_t chunk = static_cast<int64_t>((size_t(2) << 30) / bytes_per_frame);
  if (chunk < 1) chunk = 1;
  if (chunk > n_frames) chunk = n_frames;
  if (chunk > 32768) chunk = 32768;
  MDX_TRY(h->d_packed.reserve(bytes_per_frame * static_cast<size_t>(chunk)));
  MDX_TRY(h->d_wrapped.reserve(sizeof(int32_t) * static_cast<size_t>(chunk)));
  MDX_TRY(reserve_sorted_frames(h, chunk));
  for (int64_t f0 = 0; f0 < n_frames; f0 += chunk) {
    const int nf = static_cast<int>(std::min<int64_t>(chunk, n_frames - f0));
    MDX_TRY(histogram_frame_block(h, dtype, h->d_species_ptr.as<const void *>(),
                                  d_frames ? d_frames + f0 : nullptr, f0, frame_stride, nf,
                                  stream, nullptr));
  }
  if (location == MDX_HOST) {
    // staged copies are released when this function returns
    MDX_HIP(hipStreamSynchronize(stream));
  }
  return MDX_OK;
}

int mdx_rdf_read(mdx_rdf_t h, uint64_t *hist) {
  MDX_REQUIRE(h != nullptr && hist != nullptr, "null argument");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  MDX_HIP(hipMemcpyAsync(hist, h->d_hist.ptr,
                         sizeof(uint64_t) * static_cast<size_t>(h->n_keys) * h->params.nbins,
                         hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

int mdx_rdf_reduce(mdx_rdf_t h, mdx_comm_t comm) {
  MDX_REQUIRE(h != nullptr && comm != nullptr, "null argument");
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  MDX_TRY(comm_allreduce_device(comm, h->d_hist.ptr,
                                static_cast<int64_t>(h->n_keys) * h->params.nbins,
                                MDX_COMM_U64, MDX_COMM_SUM, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

}  // extern "C"
