/*
 * mdx.h -- C-ABI of libmdx, the MI355X (gfx950) trajectory-analysis kernels.
 *
 * The reference package (MDSuite) has no FFI: its calculators call TensorFlow /
 * TensorFlow-Probability ops from Python.  Each entry point below replaces the
 * numerical body of one reference function; the comment above it names that
 * function (paths relative to the reference checkout).  Everything is plain C:
 * pointers, sizes, status codes.  No torch / numpy types cross this boundary.
 *
 * Conventions
 *   - every function returns MDX_OK (0) or a negative MDX_ERR_* code; the text of
 *     the last error on the calling thread is returned by mdx_last_error();
 *   - HIP errors never abort the process, they come back as MDX_ERR_DEVICE;
 *   - "location" arguments say where a data pointer lives: MDX_HOST (the library
 *     stages it through device memory itself) or MDX_DEVICE (a pointer obtained
 *     from mdx_malloc, or any other hipMalloc'ed pointer on the current device);
 *   - the caller owns every buffer it passes in for the duration of the call;
 *     the library owns device memory behind opaque handles;
 *   - trajectory arrays use the reference's in-memory layout
 *     [n_atoms][n_frames][n_dims], C-contiguous ("atom-major"), as yielded by
 *     TrajectoryCalculator.get_batch_dataset
 *     (mdsuite/calculators/trajectory_calculator.py:346-363);
 *   - one handle / one device is driven by one host thread at a time; all work of
 *     a thread is issued on one HIP stream owned by the library (per device).
 */
#ifndef MDX_H
#define MDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_OK 0
#define MDX_ERR_INVALID_ARGUMENT (-1)
#define MDX_ERR_DEVICE (-2)
#define MDX_ERR_OUT_OF_MEMORY (-3)
#define MDX_ERR_STATE (-4)

#define MDX_HOST 0
#define MDX_DEVICE 1

#define MDX_F32 0
#define MDX_F64 1

/* ------------------------------------------------------------------ runtime */

/* Library version as major*10000 + minor*100 + patch. */
int mdx_version(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char *mdx_last_error(void);
/* Number of visible HIP devices. */
int mdx_device_count(int *count);
/* Select the device used by the calling thread (default 0). */
int mdx_set_device(int device);
/* Name, compute-unit count and HBM size of the current device. */
int mdx_device_info(char *name, size_t name_len, int *compute_units,
                    uint64_t *memory_bytes);
/* Block until all work issued on the library stream of the current device is done. */
int mdx_synchronize(void);

/* Device memory for callers that keep a trajectory resident in HBM. */
int mdx_malloc(void **dptr, size_t bytes);
int mdx_free(void *dptr);
int mdx_memcpy_h2d(void *dptr, const void *host, size_t bytes);
/* n_rows rows of row_bytes that lie pitch_bytes apart in host memory (a frame range of the
 * store's [n_particles][n_configurations][n_dims] arrays, mdsuite/database/
 * simulation_database.py:594-639 with a frame slice) arrive back to back at dptr. */
int mdx_memcpy_h2d_rows(void *dptr, const void *host, int64_t n_rows, size_t row_bytes,
                        size_t pitch_bytes);
int mdx_memcpy_d2h(void *host, const void *dptr, size_t bytes);
int mdx_memset(void *dptr, int byte_value, size_t bytes);
/* The library keeps the scratch blocks its calls free in a per-device cache (at
 * most 8 GiB) instead of returning them to the driver, and two 64 MiB page-locked host
 * buffers per device once a host array of 96 MiB or more has been uploaded; this
 * releases both. */
int mdx_release_cached_memory(void);
/* How often the library has called the driver's allocators (hipMalloc, hipHostMalloc) and
 * its own block cache since it was loaded, and the free memory of the current device:
 * lets a caller verify that a call path allocates nothing. */
int mdx_allocation_counters(uint64_t *driver_allocations, uint64_t *pool_requests,
                            uint64_t *device_free_bytes);

/* HIP events recorded on the library stream (the stream every kernel of this
 * library is launched on), for timing kernels from the host side. */
typedef struct mdx_event_s *mdx_event_t;
int mdx_event_create(mdx_event_t *ev);
int mdx_event_record(mdx_event_t ev);
/* Waits for `stop`, then returns the time between the two events. */
int mdx_event_elapsed_ms(mdx_event_t start, mdx_event_t stop, float *ms);
int mdx_event_destroy(mdx_event_t ev);

/* Kernel timing.  While profiling is on, every kernel launch of the slots below
 * is bracketed by HIP events on the library stream; mdx_profile_query waits for
 * the stream and returns the summed kernel time and launch count of one slot
 * since the last mdx_profile_reset. */
#define MDX_SLOT_RDF_PACK 0
#define MDX_SLOT_RDF_TILE 1
#define MDX_SLOT_ATOM_SUM 2
#define MDX_SLOT_ACF_PARTIAL 3
#define MDX_SLOT_ACF_FINISH 4
#define MDX_SLOT_MSD 5
#define MDX_SLOT_FLUX 6 /* momentum / thermal flux, energy-weighted sum, Kinaci heat current */
#define MDX_SLOT_TRANSFORM 7 /* unwrap and element-wise coordinate maps */
#define MDX_NUM_SLOTS 8
int mdx_profile_enable(int on);
int mdx_profile_reset(void);
int mdx_profile_query(int slot, double *total_ms, int64_t *launches);

/* Fill a device array with a deterministic synthetic trajectory (bench only):
 * element i gets lo + (hi-lo)*u(seed, i), u = 24-bit counter hash in [0,1). */
int mdx_fill_uniform_f32(float *dptr, uint64_t n, uint64_t seed, float lo, float hi);

/* Fill a device array [n_atoms][n_frames][k] with synthetic Langevin velocities (bench
 * only): every (atom, component) is an independent Ornstein-Uhlenbeck series
 * v[t+1] = decay * v[t] + kick * xi with standard normal xi, started from its stationary
 * distribution; the discretisation of the reference's own Langevin test
 * (CI/integration_tests/calculators/test_green_kubo_self_diffusion_coefficients.py:53-73:
 * decay = 1 - gamma dt / m, kick = sqrt(2 kT gamma dt) / m). */
int mdx_fill_ou_f32(float *dptr, int64_t n_atoms, int64_t n_frames, int k, uint64_t seed,
                    float decay, float kick);

/* ---------------------------------------------------------------------- RDF */
/*
 * Replaces the per-batch body of RadialDistributionFunction.run_calculator
 * (mdsuite/calculators/radial_distribution_function.py:846-885): for every
 * sampled frame, all pairs i<j of the species-concatenated atom list
 * (get_partial_triu_indices, mdsuite/utils/linalg.py:102-122), minimum-image
 * distance in float32 (get_dij :647-689, apply_minimum_image linalg.py:84-99),
 * species-pair masks and cutoff (bin_minibatch :616-645, apply_system_cutoff
 * linalg.py:125-136) and the fixed-width histogram of tf.histogram_fixed_width
 * (CPU kernel rule: bin = min(int(double(d) / (double(cutoff - 0) / nbins)),
 * nbins - 1)).  Counts are integers and bit-exact with that sequence of float32
 * operations.
 */
typedef struct mdx_rdf_s *mdx_rdf_t;

/* Reproduce the reference's strict species masks, which drop the first atom of
 * every species from both pair roles (radial_distribution_function.py:637-638).
 * Without this flag every atom takes part (the physically intended result). */
#define MDX_RDF_STRICT_REFERENCE_MASKS 1u
/* Use the kernel that performs the reference's float operations literally
 * (IEEE divide, rint, sqrt, float64 bin divide) instead of the table-driven
 * kernel that is proven equal to it; for validation. */
#define MDX_RDF_LITERAL_KERNEL 2u
/* Use the table-driven kernel (exact float32 squared distance, bin from thresholds) for
 * every frame instead of the faster kernel that evaluates an error-bounded approximation
 * and resolves the pairs near a bin edge literally; for validation. */
#define MDX_RDF_TABLE_KERNEL 4u

/*
 * n_per_species[s] atoms of species s, in the order the calculator concatenates
 * them (_format_data :535-563).  box/cutoff are float32 exactly as the reference
 * casts them (:450, :520-522).  Histogram keys are the species pairs (a,b), a<=b,
 * in itertools.combinations_with_replacement order (:272-275).
 */
int mdx_rdf_create(mdx_rdf_t *handle, int n_species, const int64_t *n_per_species,
                   const float box[3], float cutoff, int nbins, uint32_t flags);
int mdx_rdf_destroy(mdx_rdf_t handle);
/* Zero the accumulated histograms. */
int mdx_rdf_reset(mdx_rdf_t handle);
/* Number of species-pair keys = n_species*(n_species+1)/2. */
int mdx_rdf_num_keys(mdx_rdf_t handle, int *n_keys);
/* Number of atom-pair tiles one frame is cut into (the unit sharded over GPUs). */
int mdx_rdf_num_tiles(mdx_rdf_t handle, int64_t *n_tiles);
/*
 * Add the pair histogram of `n_frames` frames.  species_pos[s] points at the
 * positions of species s, [n_per_species[s]][frame_stride][3] of `dtype`
 * (float64 input is rounded to float32 first, as tf.cast does, :561-563).
 * frames[i] (host array, may be NULL for 0..n_frames-1) selects the frame index
 * along the second axis, like the reference's loop_array
 * (mdsuite/database/data_manager.py:195-203).  Only atom-pair tiles t with
 * t % shard_count == shard_rank are evaluated, so that several GPUs can split
 * one frame; pass 0, 1 for everything.  Asynchronous on the library stream.
 */
int mdx_rdf_accumulate(mdx_rdf_t handle, const void *const *species_pos, int dtype,
                       int location, int64_t frame_stride, const int64_t *frames,
                       int64_t n_frames, int shard_rank, int shard_count);
/* Copy the histograms [n_keys][nbins] (uint64 counts) to the host; synchronises. */
int mdx_rdf_read(mdx_rdf_t handle, uint64_t *hist);
/*
 * The constants of the default pair kernel for one geometry, computed on the host (no
 * device is touched).  That kernel does not perform the float32 operations of get_dij /
 * tf.histogram_fixed_width (radial_distribution_function.py:647-689, :642-644) one by
 * one: per pair it forms
 *     m_c  = half_box[c] - | |x_j[c] - x_i[c]| - half_box[c] |          (c = 0..2)
 *     bits = float_bits( fma( v_sqrt_f32( fma(m_2, m_2, fma(m_1, m_1, m_0 * m_0)) ),
 *                             scale, magic ) )
 * with magic = 2^k + shift, so that bits & 0x7FFFFF, read as a fixed-point number with
 * frac_bits fraction bits, is the pair's bin position plus shift.  The pair is dropped when
 * bits >= keep_bits, counted in bin (bits & 0x7FFFFF) >> frac_bits otherwise, and
 * evaluated again with the literal operations when the fraction bits are below
 * edge_units.  The counts equal the literal ones as long as the position error stays
 * below shift = edge_units / 2 grid units; this entry point lets a host-side model check
 * that bound against the library's own constants.  usable == 0: the geometry does not
 * admit the kernel (too many bins for the mantissa, degenerate box) and the table kernel
 * runs instead.  step and s_cut are the literal rule's bin width and the smallest float32
 * squared distance that fails d < cutoff.
 */
typedef struct {
  int usable;
  int frac_bits;
  uint32_t edge_units;
  uint32_t keep_bits;
  float magic;
  float scale;
  float s_cut;
  float half_box[3];
  double step;
} mdx_rdf_fast_plan_t;
int mdx_rdf_fast_plan(const float box[3], float cutoff, int nbins, mdx_rdf_fast_plan_t *plan);

/* --------------------------------------------------------- multi-GPU exchange */
/*
 * The reference runs in one process on one device (its only device notion is
 * mdsuite/utils/meta_functions.py:49-51); nothing of it is replaced here.  This is the
 * one exchange step of the sharded calculators: one process per GPU, every rank
 * histograms its atom-pair tiles (or frames), correlates its window origins, and the
 * partial results are summed once per run by an RCCL all-reduce over xGMI.
 *
 * Rank 0 obtains a unique id (MDX_COMM_ID_BYTES opaque bytes) and hands it to the
 * other ranks by any means; every rank then calls mdx_comm_create with the same id on
 * the device it has selected with mdx_set_device.  librccl.so is opened at the first of
 * these calls; mdx_comm_available returns 0 when that is not possible.
 */
typedef struct mdx_comm_s *mdx_comm_t;
#define MDX_COMM_ID_BYTES 128
#define MDX_COMM_U64 0
#define MDX_COMM_I64 1
#define MDX_COMM_F64 2
#define MDX_COMM_SUM 0
#define MDX_COMM_MAX 1
int mdx_comm_available(void);
int mdx_comm_unique_id(void *id, size_t bytes);
int mdx_comm_create(mdx_comm_t *comm, int rank, int size, const void *id, size_t bytes);
int mdx_comm_destroy(mdx_comm_t comm);
int mdx_comm_rank(mdx_comm_t comm, int *rank, int *size);
/* In-place all-reduce of `count` elements at `data` (MDX_HOST: staged through device
 * memory; MDX_DEVICE: reduced where it lies).  Returns when the result is in place. */
int mdx_comm_allreduce(mdx_comm_t comm, void *data, int location, int64_t count, int dtype,
                       int op);
/* Sum the accumulated histograms of all ranks in place on the device (uint64, exact in
 * any order): afterwards mdx_rdf_read returns the total on every rank.  Accumulating
 * further into a reduced handle would count the other ranks' share again. */
int mdx_rdf_reduce(mdx_rdf_t handle, mdx_comm_t comm);

/* -------------------------------------------------------- atom reductions */
/*
 * out[t][c] (+)= sum_a w_a * x[a][t][c], accumulated in float64.
 * Replaces IonicCurrent.transform_batch (mdsuite/transformations/ionic_current.py:48-58)
 * and TranslationalDipoleMoment.transform_batch
 * (mdsuite/transformations/translational_dipole_moment.py:52-62) for one species:
 * x is [n_atoms][n_frames][k]; weights is NULL (every w_a = scalar_weight, the
 * per-species charge of transformations.py:406-414) or a host array of n_atoms
 * float64 multiplied by scalar_weight.  accumulate != 0 adds to `out` (the
 * tf.add_n over species), else overwrites.  out is [n_frames][k] float64.
 * With x and out both on the device and no weight array the call returns when the work is
 * queued on the library's stream; every later library call is ordered behind it, and
 * mdx_synchronize, or any call that returns data to the host, waits for it.
 */
int mdx_weighted_atom_sum(const void *x, int dtype, int x_location, int64_t n_atoms,
                          int64_t n_frames, int k, const double *weights,
                          double scalar_weight, double *out, int out_location,
                          int accumulate);
/*
 * mdx_weighted_atom_sum for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  x ([n_atoms][n_frames][k]) and out ([n_frames][k] float64) are device
 * pointers, every atom has the weight scalar_weight, and the partial sums of the atom
 * chunks live in a static 32 MiB device array of the library (one per GPU, held until
 * the process ends) instead of an allocated block: the call makes no hipMalloc / hipFree
 * and returns MDX_ERR_INVALID_ARGUMENT when n_frames * k exceeds 4 194 304 values.  The
 * first call on a device looks up the array's address (and creates the library stream if
 * no earlier call has).  Asynchronous on the library stream; every call on one GPU uses
 * the same array, so calls that overlap on one GPU (from several host threads) race.
 */
int mdx_weighted_atom_sum_static(const void *x, int dtype, int64_t n_atoms, int64_t n_frames,
                                 int k, double scalar_weight, double *out, int accumulate);
/*
 * out[t][c] (+)= sum_a stress[a][t][3+c], c = 0..2: MomentumFlux.transform_batch
 * (mdsuite/transformations/momentum_flux.py:45-55).  stress is [n_atoms][n_frames][6].
 */
int mdx_momentum_flux(const void *stress, int dtype, int location, int64_t n_atoms,
                      int64_t n_frames, double *out, int out_location,
                      int accumulate);
/*
 * out[t][:] (+)= sum_a ((ke+pe)[a][t] * v[a][t][:] - S[a][t] . v[a][t]) with the
 * symmetric stress in the order xx,yy,zz,xy,xz,yz: ThermalFlux.transform_batch
 * (mdsuite/transformations/thermal_flux.py:51-92).  ke, pe are [n_atoms][n_frames][1].
 */
int mdx_thermal_flux(const void *stress, const void *vel, const void *ke,
                     const void *pe, int dtype, int location, int64_t n_atoms,
                     int64_t n_frames, double *out, int out_location,
                     int accumulate);

/*
 * out[t][:] (+)= sum_a x[a][t][:] * (ke[a][t] + pe[a][t]): IntegratedHeatCurrent.
 * transform_batch (mdsuite/transformations/integrated_heat_current.py:35-60), x the
 * unwrapped positions.  ke, pe are [n_atoms][n_frames][1].
 */
int mdx_energy_weighted_sum(const void *x, const void *ke, const void *pe, int dtype,
                            int location, int64_t n_atoms, int64_t n_frames, double *out,
                            int out_location, int accumulate);

/*
 * out[t][:] (+)= sum_a pos[a][t][:] * (I_a(t) + pe[a][t]) with the running integral
 * I_a(t) = carry[a] + dt * sum_{u <= t} force[a][u] . vel[a][u], one species and one batch
 * of KinaciIntegratedHeatCurrent.transform_batch (mdsuite/transformations/
 * kinaci_integrated_heat_current.py:53-87; dt = time_step * sample_rate, the cumulative
 * sum including frame t as tf.cumsum does).  carry ([n_atoms], host) holds I_a at the last
 * frame of the previous batch: read when have_carry != 0, always written with I_a at this
 * batch's last frame.  pos are the unwrapped positions; pe is [n_atoms][n_frames][1].
 * The integral belongs to an atom; the reference text sums the integrals of all species
 * element by element (which only runs for species of equal size), this entry point does not.
 */
int mdx_kinaci_heat_current(const void *pos, const void *vel, const void *force,
                            const void *pe, int dtype, int location, int64_t n_atoms,
                            int64_t n_frames, double dt, int have_carry, double *carry,
                            double *out, int out_location, int accumulate);

/* ------------------------------------------- angular distribution function */
/*
 * Weighted angle histograms of all atom triples, one batch of configurations of
 * AngularDistributionFunction._build_histograms (mdsuite/calculators/
 * angular_distribution_function.py:385-425 with get_neighbour_list / get_triplets,
 * mdsuite/utils/neighbour_list.py:54-177, and get_angles, mdsuite/utils/linalg.py:30-81).
 * Atoms are numbered through the species in the given order.  For every frame, every
 * atom i and every ordered pair (j, k), j != k, of atoms whose float32 minimum-image
 * distance to i, rounded to float16, is non-zero and below float16(cutoff):
 *   angle  = acos(clamp(u_ij . u_ik, -1, 1)),  u = r / |r| in float32
 *   weight = 1 / (|r_ij| * |r_ik|) ^ norm_power          (float32)
 * is added (float64) to hist[c][bin] when the species of (i, j, k) are the c-th
 * combination with replacement of the species list (i.e. non-decreasing species
 * order; other orders are not counted, as in the reference); bin is the one with
 * bin_edges[bin] <= angle < bin_edges[bin+1], the last bin closed, bin_edges having
 * nbins + 1 entries.  triple_counts[c] receives the number of triples.  Both outputs
 * are added to; density normalisation per batch is left to the caller.
 * species_pos / dtype / location / frame_stride / frames as in mdx_rdf_accumulate.
 */
int mdx_adf_num_combinations(int n_species);
int mdx_adf_accumulate(const void *const *species_pos, int dtype, int location,
                       int n_species, const int64_t *n_per_species, int64_t frame_stride,
                       const int64_t *frames, int64_t n_frames, const float box[3],
                       float cutoff, int nbins, const double *bin_edges, float norm_power,
                       double *hist, uint64_t *triple_counts);

/* ------------------------------------------- coordinate transformations */
/*
 * Per-atom transformations of [n_atoms][n_frames][3] arrays into arrays of the same
 * shape.  Arithmetic is float64 in the reference's operation order; out_dtype
 * MDX_F32 rounds once at the store (the database stores float32).  Input and
 * output may each live on the host or the device.
 *
 * CoordinateUnwrapper.transform_batch (mdsuite/transformations/
 * unwrap_coordinates.py:51-81): image[t] = last_image - sum_{u<=t} round((pos[u] -
 * pos[u-1]) / box) with round-half-even, out = pos + image * box.  last_pos and
 * last_image ([n_atoms][3], host) carry the state across batches: read when
 * have_carry != 0, always written with the batch's last frame.
 */
int mdx_unwrap_coordinates(const void *pos, int dtype, int location, int64_t n_atoms,
                           int64_t n_frames, const double box[3], int have_carry,
                           double *last_pos, double *last_image, void *out,
                           int out_dtype, int out_location);
/*
 * Element-wise maps:
 *   MDX_MAP_WRAP            x - floor(x / box) * box              CoordinateWrapper(center_box=False)
 *   MDX_MAP_WRAP_CENTERED   the same on x + box/2, minus box/2    CoordinateWrapper(center_box=True)
 *                           (mdsuite/transformations/wrap_coordinates.py:60-74)
 *   MDX_MAP_SCALE           x * box      ScaleCoordinates (scale_coordinates.py:47-54)
 *   MDX_MAP_UNWRAP_INDICES  x + aux * box, aux = image indices
 *                           UnwrapViaIndices (unwrap_via_indices.py:48-57)
 *   MDX_MAP_VELOCITY        (x[t+1] - x[t]) / params[0], last frame = the one before
 *                           VelocityFromPositions (velocity_from_positions.py:42-58)
 * params = box lengths, or {dt, unused, unused} for the velocity.  aux has the
 * dtype and location of x and is read by MDX_MAP_UNWRAP_INDICES only.  With input and
 * output on the device the call returns when the work is queued on the library's stream
 * (see mdx_weighted_atom_sum).
 */
#define MDX_MAP_WRAP 0
#define MDX_MAP_WRAP_CENTERED 1
#define MDX_MAP_SCALE 2
#define MDX_MAP_UNWRAP_INDICES 3
#define MDX_MAP_VELOCITY 4
int mdx_map_coordinates(int op, const void *x, const void *aux, int dtype, int location,
                        int64_t n_atoms, int64_t n_frames, const double params[3],
                        void *out, int out_dtype, int out_location);

/* ------------------------------------------------- time-lag correlations */
/*
 * Windowed autocorrelation, the ensemble loop of the Green-Kubo calculators:
 * for every window w in [win_begin, win_end) of one loaded batch
 * (dictionary_generator, mdsuite/database/data_manager.py:309-339: window w is
 * frames w*correlation_time ... + data_range), every series s and dim d:
 *     y[i]   = x[s][w*correlation_time + tau[i]][d]           i = 0..n_tau-1
 *     R[m]   = (1/(n_tau-m)) * sum_{n=0}^{n_tau-1-m} y[n]*y[n+m]
 * which is tfp.stats.auto_correlation(normalize=False, center=False) as called
 * in GreenKuboIonicConductivity.ensemble_operation
 * (mdsuite/calculators/green_kubo_ionic_conductivity.py:188-206) and
 * GreenKuboDiffusionCoefficients.ensemble_operation
 * (mdsuite/calculators/green_kubo_self_diffusion_coefficients.py:179-206).
 * Per window  c_w[m] = scale * sum_s sum_d R[m].
 * Outputs (host arrays, all float64):
 *   acf_sum[n_tau]        += sum_w c_w                     (the calculator's acf_array)
 *   If time != NULL, the per-window running integrals
 *       I_w = cumulative_trapezoid(c_w * integral_scale, x=time)  (n_tau-1 values)
 *   are folded into streaming moments over windows:
 *   integral_mean[n_tau-1], integral_m2[n_tau-1] (sum of squared deviations) of
 *   the windows of THIS call; *n_windows = win_end - win_begin.  The caller merges
 *   calls with the parallel-variance formula to obtain np.std(self.sigmas, axis=0).
 * x is [n_series][n_frames][k]; float32 input is widened to float64 exactly.
 */
int mdx_acf_windows(const void *x, int dtype, int location, int64_t n_series,
                    int64_t n_frames, int k, int64_t correlation_time,
                    int64_t win_begin, int64_t win_end, const int32_t *tau,
                    int n_tau, double scale, double integral_scale,
                    const double *time, double *acf_sum, double *integral_mean,
                    double *integral_m2, int64_t *n_windows);
/*
 * Windowed cross-correlation minus a self term, the ensemble loop of
 * GreenKuboDistinctDiffusionCoefficients (mdsuite/calculators/
 * green_kubo_distinct_diffusion_coefficients.py:176-267 with correlate,
 * mdsuite/utils/calculator_helper_methods.py:110-148).  The reference averages
 * jnp.correlate(a_i, b_j, "full")[D-1:] over all atom pairs (i, j); the correlation
 * is bilinear, so that average equals the correlation of the two species-mean
 * series a, b ([n_frames][k]).  For every window w (origin o = w * correlation_time)
 * and lag m < data_range, without the 1/(D-m) factor of the autocorrelation:
 *   c_w[m] = cross_weight * sum_{n < D-m} sum_d a[o+n+m][d] * b[o+n][d]
 *          - self_weight  * sum_s sum_{n < D-m} sum_d x_s[o+n+m][d] * x_s[o+n][d]
 * with x = self_atoms [n_self_atoms][n_frames][k] (NULL: no self term).
 *   cf_sum[m]       += sum_w c_w[m]
 *   integral_mean/m2 = mean and M2 over windows of the running trapezoid integral
 *                      of c_w over `time`, as in mdx_acf_windows (time NULL: skipped)
 * Accumulation is float64 (the reference's JAX default is float32).
 */
int mdx_ccf_windows(const void *a, const void *b, const void *self_atoms, int dtype,
                    int location, int64_t n_self_atoms, int64_t n_frames, int k,
                    int64_t correlation_time, int64_t win_begin, int64_t win_end,
                    int data_range, double cross_weight, double self_weight,
                    const double *time, double *cf_sum, double *integral_mean,
                    double *integral_m2, int64_t *n_windows);

/*
 * Windowed mean-square displacement, the ensemble loop of the Einstein
 * calculators: msd_sum[i] += scale * sum_w sum_s sum_d
 *      (x[s][w*ct + tau[i]][d] - x[s][w*ct][d])^2
 * EinsteinDiffusionCoefficients.ensemble_operation
 * (mdsuite/calculators/einstein_diffusion_coefficients.py:168-190) and
 * EinsteinHelfandIonicConductivity.ensemble_operation
 * (mdsuite/calculators/einstein_helfand_ionic_conductivity.py:165-181).
 */
int mdx_msd_windows(const void *x, int dtype, int location, int64_t n_series,
                    int64_t n_frames, int k, int64_t correlation_time,
                    int64_t win_begin, int64_t win_end, const int32_t *tau,
                    int n_tau, double scale, double *msd_sum);

#ifdef __cplusplus
}
#endif
#endif /* MDX_H */
