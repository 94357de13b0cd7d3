/*
 * C restatement of the reference's hot path for CPU timing and cross-checks.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py; the product library never links or calls it.
 *
 * Same pinning status as oracle/rdf.py and oracle/correlation.py: the RDF part is
 * PARITY UNPINNED (no golden vector exists in the reference); it is checked against
 * the NumPy restatement, which follows the reference source op by op.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/build.py).
 * -ffp-contract=off keeps every float32 multiply rounded before the next add, as in
 * the reference's separate TensorFlow ops.
 *
 * The pair loop visits the same pairs as get_partial_triu_indices
 * (mdsuite/utils/linalg.py:102-122) and applies, per pair, the float32 operations of
 * get_dij / apply_minimum_image (mdsuite/calculators/radial_distribution_function.py:647-689,
 * mdsuite/utils/linalg.py:84-99), the strict species masks and the cutoff of
 * bin_minibatch (:616-645) and the bin rule of tf.histogram_fixed_width's CPU kernel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static inline int species_of(int64_t atom, int n_species, const int64_t *first) {
  int s = 0;
  while (s + 1 < n_species && atom >= first[s + 1]) ++s;
  return s;
}

/*
 * pos: [n_atoms][frame_stride][3] float32, species concatenated in order.
 * frames: n_sel frame indices.  hist: [n_keys][nbins] uint64, ADDED to.
 * strict != 0 reproduces the reference's strict masks (first atom of every
 * species dropped from both roles).
 */
int oracle_rdf_hist(const float *pos, int n_species, const int64_t *n_per_species,
                    int64_t frame_stride, const int64_t *frames, int64_t n_sel,
                    const float *box, float cutoff, int nbins, int strict,
                    uint64_t *hist) {
  int64_t first[1025];
  int64_t n_atoms = 0;
  if (n_species < 1 || n_species > 1024) return -1;
  for (int s = 0; s < n_species; ++s) {
    first[s] = n_atoms;
    n_atoms += n_per_species[s];
  }
  first[n_species] = n_atoms;
  const int n_keys = n_species * (n_species + 1) / 2;
  /* key index of (a,b), a<=b, in combinations_with_replacement order */
  int *key_of = (int *)malloc(sizeof(int) * n_species * n_species);
  int *spec = (int *)malloc(sizeof(int) * (n_atoms > 0 ? n_atoms : 1));
  unsigned char *valid = (unsigned char *)malloc(n_atoms > 0 ? n_atoms : 1);
  if (!key_of || !spec || !valid) return -2;
  {
    int key = 0;
    for (int a = 0; a < n_species; ++a)
      for (int b = a; b < n_species; ++b) key_of[a * n_species + b] = key++;
  }
  for (int64_t i = 0; i < n_atoms; ++i) {
    spec[i] = species_of(i, n_species, first);
    /* strict masks: start < index < stop drops index == start */
    valid[i] = (unsigned char)(!strict || i != first[spec[i]]);
  }
  const float width = cutoff - 0.0f;
  const double step = (double)width / (double)nbins;
  const double top = (double)(nbins - 1);
  const float bx = box[0], by = box[1], bz = box[2];
  const size_t hist_len = (size_t)n_keys * nbins;

#pragma omp parallel
  {
    uint64_t *local = (uint64_t *)calloc(hist_len, sizeof(uint64_t));
    for (int64_t f = 0; f < n_sel; ++f) {
      const int64_t frame = frames ? frames[f] : f;
#pragma omp for schedule(dynamic, 16) nowait
      for (int64_t i = 0; i < n_atoms; ++i) {
        if (!valid[i]) continue;
        const float *pi = pos + (i * frame_stride + frame) * 3;
        const float xi = pi[0], yi = pi[1], zi = pi[2];
        const int a = spec[i];
        for (int64_t j = i + 1; j < n_atoms; ++j) {
          if (!valid[j]) continue;
          const float *pj = pos + (j * frame_stride + frame) * 3;
          float rx = pj[0] - xi;
          float ry = pj[1] - yi;
          float rz = pj[2] - zi;
          rx = rx - rintf(rx / bx) * bx;
          ry = ry - rintf(ry / by) * by;
          rz = rz - rintf(rz / bz) * bz;
          const float s = (rx * rx + ry * ry) + rz * rz;
          const float d = sqrtf(s);
          if (d < cutoff) {
            double q = (double)d / step;
            if (q > top) q = top;
            const int bin = (int)q;
            local[(size_t)key_of[a * n_species + spec[j]] * nbins + bin] += 1u;
          }
        }
      }
    }
#pragma omp critical
    {
      for (size_t e = 0; e < hist_len; ++e) hist[e] += local[e];
    }
    free(local);
  }
  free(key_of);
  free(spec);
  free(valid);
  return 0;
}

/* out[t][c] (+)= sum_a w * x[a][t][c]; x float32 [n_atoms][n_frames][k]
 * (IonicCurrent.transform_batch, mdsuite/transformations/ionic_current.py:48-58). */
int oracle_weighted_atom_sum_f32(const float *x, int64_t n_atoms, int64_t n_frames, int k,
                                 double weight, double *out, int accumulate) {
  const int64_t row = n_frames * k;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < row; ++c) {
    double acc = 0.0;
    for (int64_t a = 0; a < n_atoms; ++a) acc = acc + weight * (double)x[a * row + c];
    out[c] = accumulate ? out[c] + acc : acc;
  }
  return 0;
}

/*
 * Windowed autocorrelation by direct float64 sums, same estimator as
 * tfp.stats.auto_correlation(normalize=False, center=False): lag m divided by
 * (R - m).  x: float64 [n_series][n_frames][k].  acf_sum[R] is ADDED to;
 * per_window (may be NULL) receives c_w[m] for every window: [n_win][R].
 * (GreenKuboIonicConductivity.ensemble_operation,
 *  mdsuite/calculators/green_kubo_ionic_conductivity.py:188-206.)
 */
int oracle_acf_windows(const double *x, int64_t n_series, int64_t n_frames, int k,
                       int64_t ct, int64_t win_begin, int64_t win_end, const int32_t *tau,
                       int R, double scale, double *acf_sum, double *per_window) {
  const int64_t n_win = win_end - win_begin;
  double *tmp = per_window;
  if (!tmp) tmp = (double *)calloc((size_t)(n_win > 0 ? n_win : 1) * R, sizeof(double));
  if (!tmp) return -2;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t w = 0; w < n_win; ++w) {
    const int64_t f0 = (win_begin + w) * ct;
    for (int m = 0; m < R; ++m) {
      double total = 0.0;
      for (int64_t s = 0; s < n_series; ++s) {
        const double *xs = x + s * n_frames * k;
        for (int d = 0; d < k; ++d) {
          double acc = 0.0;
          for (int n = 0; n + m < R; ++n)
            acc = acc + xs[(f0 + tau[n]) * k + d] * xs[(f0 + tau[n + m]) * k + d];
          total = total + acc / (double)(R - m);
        }
      }
      tmp[w * R + m] = scale * total;
    }
  }
  for (int64_t w = 0; w < n_win; ++w)
    for (int m = 0; m < R; ++m) acf_sum[m] += tmp[w * R + m];
  if (!per_window) free(tmp);
  return 0;
}

/* msd_sum[i] += scale * sum_w sum_s sum_d (x[s][w*ct+tau[i]][d] - x[s][w*ct][d])^2
 * (EinsteinDiffusionCoefficients.ensemble_operation,
 *  mdsuite/calculators/einstein_diffusion_coefficients.py:168-190). */
int oracle_msd_windows(const double *x, int64_t n_series, int64_t n_frames, int k,
                       int64_t ct, int64_t win_begin, int64_t win_end, const int32_t *tau,
                       int R, double scale, double *msd_sum) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < R; ++i) {
    double acc = 0.0;
    for (int64_t w = win_begin; w < win_end; ++w) {
      const int64_t f0 = w * ct;
      for (int64_t s = 0; s < n_series; ++s) {
        const double *xs = x + s * n_frames * k;
        for (int d = 0; d < k; ++d) {
          const double diff = xs[(f0 + tau[i]) * k + d] - xs[f0 * k + d];
          acc = acc + diff * diff;
        }
      }
    }
    msd_sum[i] += scale * acc;
  }
  return 0;
}
