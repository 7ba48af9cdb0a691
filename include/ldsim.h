/*
 * ldsim.h -- C-ABI of libldsim_hip.so: larnd-sim's charge/light hot path on MI355X (gfx950).
 *
 * The reference (DUNE/larnd-sim @2024-10-16) has no FFI for this path: the path sits behind
 * Python call sites `module.kernel[blocks, threads](ndarray, ...)` in one driver loop
 * (reference cli/simulate_pixels.py:732,742,797,923,944,1002,1016,1036,1057,1087,1150) plus
 * module-global constants frozen at JIT time.  Each entry point below names the reference
 * kernel / call site it replaces.  Conventions (same as the reference's call sites):
 *   - the caller owns every buffer; outputs are sized by the caller;
 *   - `tracks` is an array of records (any layout described by LdsimTrackLayout) mutated in place;
 *   - every function returns 0 on success or a negative LDSIM_E* code; ldsim_last_error()
 *     returns a thread-local message.  Nothing throws across the boundary.
 *   - one ldsim_ctx per device/process; a ctx is thread-compatible, not thread-safe.
 *
 * Two families:
 *   (1) stage-by-stage, HOST buffers in / out ("materialising" forms: one per reference kernel,
 *       used by the parity tests and by a drop-in driver);
 *   (2) device-resident chain: upload the segments once, run quench -> drift -> pixels ->
 *       induced current -> per-pixel sum -> ADC entirely in HBM, download per-pixel results.
 */
#ifndef LDSIM_H
#define LDSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDSIM_ABI_VERSION 8

/* error codes */
#define LDSIM_OK 0
#define LDSIM_EINVAL (-1)   /* bad argument */
#define LDSIM_EHIP (-2)     /* HIP runtime error (message has the hipError string) */
#define LDSIM_ENOSPC (-3)   /* caller-provided capacity too small (required size reported) */
#define LDSIM_ESTATE (-4)   /* call order violated (e.g. no response table set) */
#define LDSIM_ENODEV (-5)   /* no usable GPU */

/* record field dtype codes */
enum { LDSIM_F4 = 1, LDSIM_F8 = 2, LDSIM_I4 = 3, LDSIM_U4 = 4, LDSIM_I8 = 5, LDSIM_U8 = 6 };

/* hot-path record fields (reference cli/dumpTree.py:17-28 names) */
enum ldsim_field {
  LDSIM_X_START = 0, LDSIM_Y_START, LDSIM_Z_START, LDSIM_X_END, LDSIM_Y_END, LDSIM_Z_END,
  LDSIM_X, LDSIM_Y, LDSIM_Z, LDSIM_DEDX, LDSIM_DE, LDSIM_T, LDSIM_T_START, LDSIM_T_END,
  LDSIM_T0, LDSIM_T0_START, LDSIM_T0_END, LDSIM_N_ELECTRONS, LDSIM_N_PHOTONS,
  LDSIM_LONG_DIFF, LDSIM_TRAN_DIFF, LDSIM_PIXEL_PLANE, LDSIM_NFIELDS
};

/* Where each field lives inside one record; offset < 0 = field absent (reads 0, never written). */
typedef struct {
  int32_t itemsize;
  int32_t offset[LDSIM_NFIELDS];
  int32_t dtype[LDSIM_NFIELDS];
} LdsimTrackLayout;

#define LDSIM_MAX_TPC 128

/* Constants the reference keeps as module globals (larndsim/consts/{detector,light,sim,physics}.py),
 * passed as a plain struct instead of being frozen into JIT-compiled kernels. */
typedef struct {
  /* physics.py */
  double box_alpha, box_beta, birks_ab, birks_kb, w_ion;
  /* light.py: quenching needs W_PH and SCINT_PRESCALE */
  double w_ph, scint_prescale;
  /* detector.py drift */
  double e_field, lar_density, v_drift, electron_lifetime, long_diff, tran_diff;
  /* geometry */
  int32_t n_tpc;
  int32_t default_plane_index;             /* 0xBEEF */
  double tpc_borders[LDSIM_MAX_TPC][3][2]; /* [tpc][x,y,z][lo,hi]; anode = [.][2][0] */
  int32_t n_pixels[2];
  int32_t sampled_points;                  /* 40 */
  int32_t n_time_ticks;                    /* len(TIME_TICKS) */
  double pixel_pitch;
  /* time / response */
  double time_sampling, time_padding, time_window, time_interval[2];
  double response_sampling, response_bin_size;
  /* FEE */
  double discrimination_threshold, clock_cycle, buffer_risetime, gain, v_cm, v_ref, v_pedestal;
  int32_t adc_hold_delay, adc_busy_delay, reset_cycles, adc_counts;
  double reset_noise_charge, uncorrelated_noise_charge, discriminator_noise;
  int32_t max_tracks_per_pixel, max_adc_values;
  /* light */
  int32_t light_trig_mode, enable_lut_smearing;
  int32_t n_op_channel, max_mc_truth_ids;
  double light_tick_size, mc_truth_threshold;
  /* light waveform response (ABI 2): scintillation_model (light_sim.py:131-146), sipm_response_model (:274-300) */
  double light_window[2];                  /* LIGHT_WINDOW: conv_ticks = ceil((w[1] - w[0]) / LIGHT_TICK_SIZE) */
  double singlet_fraction, tau_s, tau_t;
  double light_response_time, light_oscillation_period, impulse_tick_size;
  int32_t sipm_response_model;             /* 0 = RLC model, 1 = measured impulse (IMPULSE_MODEL passed as an array) */
  int32_t mc_sample_multiplier;            /* sim.MC_SAMPLE_MULTIPLIER (ABI 3, tracks_current_mc, detsim.py:318-323) */
  double min_step_size;                    /* sim.MIN_STEP_SIZE [cm] */
  /* light triggers and digitisation (ABI 4): get_triggers / sim_triggers / gen_light_detector_noise, light_sim.py:339-619 */
  double light_trig_window[2];             /* LIGHT_TRIG_WINDOW [us] */
  double light_digit_sample_spacing;       /* LIGHT_DIGIT_SAMPLE_SPACING [us] */
  double light_det_noise_sample_spacing;   /* LIGHT_DET_NOISE_SAMPLE_SPACING [us] */
  int32_t light_nbit;                      /* LIGHT_NBIT */
  int32_t op_channel_per_trig;             /* OP_CHANNEL_PER_TRIG */
} LdsimConsts;

typedef struct ldsim_ctx ldsim_ctx;

/* ---- context ------------------------------------------------------------------------------ */
const char* ldsim_last_error(void);
int ldsim_abi_version(void);
int ldsim_device_count(void);
/* consts.load_properties + importlib.reload (cli/simulate_pixels.py:431,458-464) */
int ldsim_ctx_create(int device, const LdsimConsts* consts, ldsim_ctx** out);
int ldsim_ctx_destroy(ldsim_ctx* ctx);
/* page-locked host memory for the download / upload buffers of a caller that wants PCIe-rate copies (hipHostMalloc) */
int ldsim_host_alloc(void** p, size_t bytes);
int ldsim_host_free(void* p);
int ldsim_set_consts(ldsim_ctx* ctx, const LdsimConsts* consts);
/* cp.load(response_file) (cli/simulate_pixels.py:436): f64 table [ni][nj][nk], host pointer */
int ldsim_set_response(ldsim_ctx* ctx, const double* response, int32_t ni, int32_t nj, int32_t nk);
/* light.OP_CHANNEL_EFFICIENCY / OP_CHANNEL_TO_TPC (consts/light.py:109-119) */
int ldsim_set_light_channels(ldsim_ctx* ctx, const double* efficiency, const int32_t* op_channel_to_tpc, int32_t n);
/* np.load(light_lut)['arr'] as SoA planes: vis,t0,t0_avg [nx*ny*nz*ndet] f32; time_dist [..*nprof] f32 */
int ldsim_set_light_lut(ldsim_ctx* ctx, const float* vis, const float* t0, const float* t0_avg,
                        const float* time_dist, int32_t nx, int32_t ny, int32_t nz, int32_t ndet, int32_t nprof);
/* tuning / validation knobs: "prune_log" (weights below exp(-v) of the local peak are skipped, 0 = keep all;
 * default 23 = 1e-10 of the peak like the quadrature rule: ADC charges move by < 1e-8 relative against keeping everything,
 * tools/prune_sweep.py; 30 was the default before and costs 16-19 % more correlation work),
 * "tail_log" (split path: charge samples bounded by exp(-v) of the segment's peak density are evaluated in f32,
 * relative error ~3e-7 of a term that small; 0 = every sample in f64),
 * "trim_response" (1 = skip leading/trailing response ticks that are empty for every cell), "trim_response_log" (what empty
 * means: below exp(-v) of the table's largest entry; default 23 = 1e-10, the level the weights are pruned at -- such ticks move a
 * waveform by less than 1e-10 of its peak; 0 = exactly 0.0 only),
 * "split_kernels" (1 = weights_kernel + mac_kernel, 0 = monolithic current_kernel),
 * "wbuf_doubles_per_pair" (initial average budget of the split path's weight pool; the pool grows to the measured
 * demand and the launch is repeated when it was exhausted, so this only affects the first launches; setting it forgets
 * the size learned so far), "split_max_items" (validation: pairs with more weight items than this are recomputed by
 * the monolithic kernel; 0 = the built-in capacity: 512 items at TIME_SAMPLING/RESPONSE_SAMPLING = 1, 2048 at 2),
 * "weights_mode" (split path: 2 = node-separable form, gtables_kernel + gcorr_kernel: the quadrature's tables correlated on the
 * f64 matrix pipe, no weight pool ("mac_mode" does not apply); 1 = qweights_kernel, Gauss-Legendre quadrature along the
 * segment, then the correlation kernel of "mac_mode"; 0 = weights_kernel, the closed form per charge sample),
 * "gform_max_support" (weights_mode 2 runs the node-separable form for response tables whose staged support is at most this
 * many time ticks and the kernels of weights_mode 1 for wider ones: the matrix form pays per response tick, the shifted-window
 * kernels per 512-tick tile; 0 = never, 1e9 = always = the default since round 4 -- 768 before, when full-support tables were
 * faster in the shifted-window kernels), "gform_chunks" (weights_mode 2: tables and correlation in this many ranges of the pair list, the tables of a range on a second
 * stream beside the correlation of the one before; 1..32, default 1: measured no faster), "gform_wave_tables" (weights_mode 2, tables stage: 1 =
 * gtables_wave_kernel, one wave per pair, for the pairs that fit it and gtables_kernel, one workgroup per pair, for the rest
 * (default); 0 = gtables_kernel for all: same tables entry for entry), "quad_max_nodes" (qweights_kernel: pairs that need more nodes,
 * i.e. segments longer than ~value/2 Gaussian widths, are recomputed by the monolithic kernel; 8..256, default 256),
 * "quad_accuracy_log10" (tables / weights stage: Gauss-Legendre nodes along the segment for a quadrature error of 1e-v of the peak
 * weight, v = 7 (default since round 4: N = ceil(3.4 + 1.38 r) nodes for a clipped segment r Gaussian widths long; per tick the
 * result stays at 0.012 of the parity tolerance 1e-5 |ref| + 1e-7 peak against the reference's goldens, where the f4 rounding of
 * the stored currents already is -- tools/quad_sweep.py, profiles/r04_acc_sweep_module0.log), 8, 9, 10 (4.8 + 1.6 r, the default
 * of rounds 2-3), 12 (6 + 1.9 r), or looser: 6 (3.0 + 1.3 r: 3.7 % faster, 0.033 of the tolerance at worst, but the pixel charges
 * move by 4.5e-8 against the closed form -- beyond the 2e-8 the parity tests hold them to, so it is not the default) and 5
 * (2.8 + 1.2 r: 0.32 of the tolerance) -- profiles/r04_acc_sweep_module0_5_6.log),
 * "mac_mode" (split path, correlation stage: 1 = mac_shift_kernel / mac_shift2_kernel (default), 0 = mac_kernel<M>, rows
 * staged in LDS; bit-identical results), "light_incidence_scalar" (1 = calculate_light_incidence with one channel per lane
 * instead of four (the form used when n_out or the LUT's detector count is not a multiple of 4); identical bits; default 0),
 * "light_sum_no_list" (1 = ldsim_dev_sum_light without truth slots launches a workgroup per (detector, tick tile) and clears
 * the whole array first, instead of summing over the device-built list of the lit tiles; same cells, values to the order of
 * the f64 additions; default 0), "light_sum_async" (1 = the ldsim_dev_sum_light calls without truth slots launch on a stream of the
 * ctx's own beside whatever its main stream carries next, e.g. the charge chain of the same segments; every entry point that
 * writes what they read -- segment upload / reset / quench-drift, incidence, LUT, channel tables, constants -- or reads their array
 * -- ldsim_dev_light_download, ldsim_dev_light_response, ldsim_synchronize -- waits for them first; default 0),
 * "light_truth_lds" (the truth slots of ldsim_scintillation_effect / ldsim_light_detector_response / ldsim_dev_light_response, at most
 * 64 of them: 1 (default) = by light_truth_lds_kernel, a wave's 64 output rows in LDS beside the plain sum; 0 = inside light_conv_kernel
 * on slot-major copies of the rows in memory -- same bits for rows of distinct ids in front of their first -1, and the only path
 * that follows light_sim.py:331-335 literally for other rows is the default),
 * "numba_f32" (1 = the sub-expressions Numba types float32 for f4 record fields are evaluated in float, detsim.py:74-79,
 * 116-118,141,387; 0 = all-f64, what the reference computes for f8 records and what the goldens pin; default 0),
 * "mc_current" (1 = the fused chain takes its induced currents from tracks_current_mc like the reference driver does;
 * default 0 = tracks_current),
 * "debug_phases" (timing tools only, tools/phase_timing3.py: bit mask that drops phases of the tracks_current kernels;
 * results are wrong unless it is 15, the default).  An unknown name returns LDSIM_EINVAL. */
int ldsim_set_option(ldsim_ctx* ctx, const char* name, double value);
/* Per-pixel discrimination thresholds and gains of the fused chain -- the reference driver's
 * pixel_thresholds_lut[unique_pix] (cli/simulate_pixels.py:1079-1084) and pixel_gains_lut[unique_pix] (:1097-1100),
 * CudaDict lookups of util/cuda_dict.py:49-53 loaded from an .npz with keys / values / default (:82-88).
 * keys[n] are pixel ids (unique), values[n] their entries, every other pixel id gets default_value; keys outside the
 * current pixel geometry can never be looked up and are skipped.  Held as a dense table over
 * n_pixels[0]*n_pixels[1]*n_tpc ids: set the constants first, and set the tables again after a change of geometry
 * (ldsim_charge_chain returns LDSIM_ESTATE otherwise).  Without a table the chain uses DISCRIMINATION_THRESHOLD*e and
 * GAIN*mV/e.  The stage functions ldsim_get_adc_values / ldsim_digitize take their arrays from the caller as before. */
int ldsim_set_pixel_thresholds(ldsim_ctx* ctx, const int32_t* keys, const double* values, int64_t n,
                               double default_value);
int ldsim_set_pixel_gains(ldsim_ctx* ctx, const int32_t* keys, const double* values, int64_t n, double default_value);
int ldsim_clear_pixel_tables(ldsim_ctx* ctx);
int ldsim_synchronize(ldsim_ctx* ctx);

/* Random streams of the noisy stages: numba.cuda.random.create_xoroshiro128p_states(n, seed) (cli/simulate_pixels.py:
 * 92-104,396; third-party algorithm restated in csrc/rng.h -- unpinned).  State ip serves pixel row ip of
 * a get_adc_values call / of a chain launch and is advanced in place like the reference's rng_states[ip]; the table grows
 * on demand by continuing the 2^64-step jump sequence.  Needed before any call with a non-zero noise charge
 * (RESET_NOISE_CHARGE, UNCORRELATED_NOISE_CHARGE, DISCRIMINATOR_NOISE): such a call without it returns LDSIM_ESTATE. */
int ldsim_rng_seed(ldsim_ctx* ctx, uint64_t seed, int64_t n_states);
int ldsim_rng_states_download(ldsim_ctx* ctx, uint64_t* states /* [n][2] = s0, s1 */, int64_t n);
int ldsim_rng_clear(ldsim_ctx* ctx);
/* maybe_create_rng_states(n, seed, rng_states) (cli/simulate_pixels.py:92-104): no table -> n states from `seed`; a shorter
 * table -> create_xoroshiro128p_states(n - len, seed) appended; otherwise untouched.  ldsim_rng_count: states held, -1 none. */
int ldsim_rng_extend(ldsim_ctx* ctx, int64_t n_states, uint64_t seed);
int64_t ldsim_rng_count(ldsim_ctx* ctx);

/* ---- (1) stage-by-stage, host buffers ---------------------------------------------------------- */
/* quenching.quench[bpg,tpb](tracks, mode)            -- reference larndsim/quenching.py:11-44 */
int ldsim_quench(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* layout, int32_t mode);
/* drifting.drift[bpg,tpb](tracks)                    -- larndsim/drifting.py:11-58 */
int ldsim_drift(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* layout);
/* pixels_from_track.max_pixels[bpg,tpb](tracks, n_max) -- larndsim/pixels_from_track.py:43-65 */
int ldsim_max_pixels(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                     int64_t* n_max_pixels);
/* pixels_from_track.get_pixels[bpg,tpb](tracks, active, neigh, nrad, n_pixels_list, radius)
 *                                                     -- larndsim/pixels_from_track.py:67-109 */
int ldsim_get_pixels(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                     int32_t radius, int32_t* active_pixels, int32_t max_active,
                     int32_t* neighboring_pixels, int32_t* neighboring_radius, int32_t max_neigh,
                     double* n_pixels_list);
/* detsim.time_intervals[bpg,tpb](track_starts, time_max, tracks) -- larndsim/detsim.py:18-40 */
int ldsim_time_intervals(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                         double* track_starts, int64_t* time_max);
/* detsim.tracks_current[(S,P,T/64),(1,1,64)](signals, pixels, tracks, response)
 *                                                     -- larndsim/detsim.py:351-453
 * Runs the kernels the options select, exactly like ldsim_charge_chain does for its sorted pair list ("split_kernels",
 * "weights_mode", "mac_mode", "gform_max_support").  Default: the node-separable form (weights_mode 2) -- pair_setup_kernel,
 * gtables_wave_kernel (Gauss-Legendre tables X, Y, Z per pair) and gcorr_kernel (the two products on v_mfma_f64_16x16x4) -- for
 * response tables of any support (gform_max_support ticks at most, default unlimited; wider tables would be handed to round 2's
 * qweights_kernel + mac_shift kernels, weights_mode 1); pairs beyond a kernel's capacities (rules of more than quad_max_nodes
 * nodes ...) are recomputed by the monolithic closed-form kernel.  Response ticks below exp(-trim_response_log) (default 23:
 * 1e-10) of the table's largest entry are not read: an approximation relative to the reference of < 1e-10 of a waveform's peak
 * that the headline throughput on the SURVEY table relies on (bench.py reports `exact_zero_trim` beside it).  All on the dense
 * [S][P] pixel array; pixel id -1 is evaluated as the reference evaluates it (Python index wrap). */
int ldsim_tracks_current(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                         const int32_t* pixels, int32_t max_neigh, float* signals, int32_t n_ticks);
/* detsim.tracks_current_mc[(S,P,T/64),(1,1,64)](signals, pixels, tracks, response, rng_states) -- larndsim/detsim.py:258-348,
 * what the reference driver calls (cli/simulate_pixels.py:1016).  The reference lets all tick threads of a (segment, pixel)
 * race on rng_states[itrk + ntrk*ipix]; here every (segment, pixel, tick) draws from its own stream derived from that state
 * (SplitMix64 of its words and the tick index), so the result is reproducible and statistically equivalent, not equal.
 * Needs ldsim_rng_seed; the table is grown to n * max_neigh states and each used state is stepped once afterwards. */
int ldsim_tracks_current_mc(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                            const int32_t* pixels, int32_t max_neigh, float* signals, int32_t n_ticks);
/* detsim.get_track_pixel_map2[bpg,tpb](track_pixel_map, unique_pix, pixels, distances, max_distance)
 *                                                     -- larndsim/detsim.py:564-607 */
int ldsim_track_pixel_map(ldsim_ctx* ctx, const int32_t* unique_pix, int64_t n_unique,
                          const int32_t* pixels, const int32_t* distances, int64_t n, int32_t max_neigh,
                          int32_t max_distance, int64_t* track_pixel_map, int32_t max_tracks);
/* detsim.sum_pixel_signals[...](pixels_signals, signals, track_starts, pixel_index_map, track_pixel_map,
 *                               pixels_tracks_signals, overflow_flag) -- larndsim/detsim.py:468-527 */
int ldsim_sum_pixel_signals(ldsim_ctx* ctx, const float* signals, int64_t n, int32_t max_neigh, int32_t n_ticks,
                            const double* track_starts, const int64_t* pixel_index_map,
                            const int64_t* track_pixel_map, int32_t max_tracks, int64_t n_unique,
                            double* pixels_signals, double* pixels_tracks_signals /* may be NULL */,
                            double* overflow_flag);
/* fee.get_adc_values[bpg,tpb](pixels_signals, pixels_signals_tracks, time_ticks, adc_list, adc_ticks_list,
 *                             time_padding, rng_states, current_fractions, pixel_thresholds)
 *                                                     -- larndsim/fee.py:517-655; with non-zero noise charges rng_states = the
 *                                                        table of ldsim_rng_seed (state ip for pixel ip) */
int ldsim_get_adc_values(ldsim_ctx* ctx, const double* pixels_signals, const double* pixels_signals_tracks,
                         int64_t n_unique, int32_t n_ticks, int32_t max_tracks, const double* time_ticks,
                         int32_t n_time_ticks, double time_padding, const double* pixel_thresholds,
                         double* adc_list, double* adc_ticks_list, double* current_fractions /* may be NULL */);
/* fee.digitize(integral_list, gain)                   -- larndsim/fee.py:499-515; gain NULL = GAIN*mV/e */
int ldsim_digitize(ldsim_ctx* ctx, const double* integral_list, int64_t n, const double* gain_list, double* adcs);
/* lightLUT.calculate_light_incidence[bpg,tpb](tracks, lut, light_incidence, voxel)
 *                                                     -- larndsim/lightLUT.py:65-136 */
int ldsim_light_incidence(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                          int32_t n_out_channels, float* n_photons_det, float* t0_det, int32_t* voxel);
/* light_sim.sum_light_signals[...](segments, segment_voxel, segment_track_id, light_inc, op_channel, lut,
 *   start_time, light_sample_inc, true_track_id, true_photons, sorted_indices, t0_profile_length)
 *                                                     -- larndsim/light_sim.py:58-129 */
int ldsim_sum_light_signals(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                            const int32_t* voxel, const int64_t* segment_track_id,
                            const float* n_photons_det, int32_t n_inc_channels,
                            const int32_t* op_channel, int32_t n_det, const int32_t* sorted_indices,
                            double start_time, int32_t n_ticks,
                            float* light_sample_inc, int64_t* true_track_id, double* true_photons,
                            int32_t max_truth);

/* Light waveform response, the two deterministic stages after the photon sum.  Arrays are [n_det][n_ticks]
 * f4 and truth slots [n_det][n_ticks][max_truth] (i8 ids, f8 photons; max_truth may be 0 with NULL pointers).  Outputs are
 * accumulated into: the caller pre-fills them like the driver does (zeros, ids -1; cli/simulate_pixels.py:1160-1162,
 * 1172-1174).  Every term is added in ascending tick order with an f4 store after it, as the reference's kernels do.
 * light_sim.calc_scintillation_effect[bpg,tpb](light_sample_inc, true_track_id, true_photons, scint, scint_true_track_id,
 *                                              scint_true_photons)        -- larndsim/light_sim.py:148-184, model :131-146 */
int ldsim_scintillation_effect(ldsim_ctx* ctx, const float* light_sample_inc, const int64_t* true_track_id,
                               const double* true_photons, int32_t n_det, int32_t n_ticks, int32_t max_truth,
                               float* scint, int64_t* scint_true_track_id, double* scint_true_photons);
/* light_sim.calc_light_detector_response[bpg,tpb](light_sample_inc, true_track_id, true_photons, light_response,
 *                                                 response_true_track_id, response_true_photons)
 *                                                                         -- larndsim/light_sim.py:303-337, model :274-300
 * light_gain[n_det] is LIGHT_GAIN indexed by array ROW like the reference (:320); impulse_model[n_impulse] is IMPULSE_MODEL,
 * read when sipm_response_model == 1.  The truth part keeps the reference's slot test on the input ids at [idet, itick]. */
int ldsim_light_detector_response(ldsim_ctx* ctx, const float* light_sample_inc, const int64_t* true_track_id,
                                  const double* true_photons, int32_t n_det, int32_t n_ticks, int32_t max_truth,
                                  const double* light_gain, const double* impulse_model, int32_t n_impulse,
                                  float* response, int64_t* response_true_track_id, double* response_true_photons);

/* Second half of the light chain: Poisson fluctuations, triggers, detector noise, digitised waveforms.
 * light_sim.calc_stat_fluctuations[bpg,tpb](light_sample_inc, light_sample_inc_disc, rng_states)
 *                                                                         -- larndsim/light_sim.py:186-238
 * Element (idet, itick) draws from state idet*n_ticks + itick of the ldsim_rng_seed table (advanced in place): one float32
 * uniform for a mean below 30 PE per tick, one float32 normal above.  The generator is third-party and unpinned. */
int ldsim_stat_fluctuations(ldsim_ctx* ctx, const float* light_sample_inc, int32_t n_det, int32_t n_ticks,
                            float* light_sample_inc_disc);
/* light_sim.get_triggers(signal, group_threshold, op_channel_idx, i_subbatch), LIGHT_TRIG_MODE 0 branch
 *                                                                         -- larndsim/light_sim.py:339-411
 * signal [n_det][n_ticks] f4 (NULL: the resident response of ldsim_dev_light_response), group_threshold [n_grp] with
 * n_det = n_grp * OP_CHANNEL_PER_TRIG; row_module [n_det] = index (0..n_mod-1, ascending module id) of the module whose
 * channels contain the row's optical channel, -1 for none.  Returns the trigger ticks module by module in the order the
 * reference appends them, with the reference's bookkeeping of the re-sliced mask (:404-411).  LDSIM_ENOSPC with *n_trig =
 * the number found when `capacity` is too small.  (LIGHT_TRIG_MODE 1 is one trigger at tick 0 and needs no call.) */
int ldsim_light_triggers(ldsim_ctx* ctx, const float* signal, int32_t n_det, int32_t n_ticks, const double* group_threshold,
                         int32_t n_grp, const int32_t* row_module, int32_t n_mod, int64_t* trigger_idx,
                         int32_t* trigger_module, int64_t capacity, int64_t* n_trig);
/* light_sim.gen_light_detector_noise(shape, light_det_noise)              -- larndsim/light_sim.py:445-478
 * spectrum [n_rows][nbins] (the rows light_det_noise[...] selects), noise [n_rows][n_samples] f8.  phases [n_rows]
 * [n_samples/2+1] = the uniform numbers of :465, or NULL to draw them from a counter hash seeded by ldsim_rng_seed (the
 * reference draws them from cupy's global generator: unpinned by construction).  n_samples < 2 is refused (NaN there). */
int ldsim_light_detector_noise(ldsim_ctx* ctx, int32_t n_rows, int32_t n_samples, const double* spectrum, int32_t nbins,
                               const double* phases, double* noise);
/* light_sim.sim_triggers(bpg, tpb, signal, signal_op_channel_idx, signal_true_track_id, signal_true_photons, trigger_idx,
 *                        op_channel_idx, digit_samples, light_det_noise) incl. digitize_signal
 *                                                                         -- larndsim/light_sim.py:480-619
 * signal [n_det][n_ticks] f4 with truth [n_det][n_ticks][max_truth] (signal NULL: the resident response and its truth);
 * trigger_op_channel_idx [n_trig][n_det_trig]; light_det_noise [n_noise_channels][n_noise_bins] indexed by optical channel
 * (NULL or all zero: no noise).  phases_signal [n_det][Tp/2+1] / phases_missing [n_missing][Tp/2+1] (Tp = padded length)
 * stand for the two cp.random.uniform calls, NULL = counter hash.  Outputs digit_signal [n_trig][n_det_trig][digit_samples]
 * f8 (rounded to the digitiser's LSB), truth ids (i8, -1) / photons (f8) [..][max_truth].  The padded copies of the
 * reference are not materialised; an f4 signal that needs no padding keeps its f4 rounding when the noise is added. */
int ldsim_sim_triggers(ldsim_ctx* ctx, const float* signal, const int32_t* signal_op_channel_idx, int32_t n_det,
                       int32_t n_ticks, const int64_t* signal_true_track_id, const double* signal_true_photons,
                       int32_t max_truth, const int64_t* trigger_idx, int32_t n_trig,
                       const int32_t* trigger_op_channel_idx, int32_t n_det_trig, int32_t digit_samples,
                       const double* light_det_noise, int32_t n_noise_channels, int32_t n_noise_bins,
                       const double* phases_signal, const double* phases_missing, double* digit_signal,
                       int64_t* digit_true_track_id, double* digit_true_photons);

/* ---- (2) device-resident chain ---------------------------------------------------------------------- */
/* Upload `n` records (H2D) and unpack them into the ctx's SoA segment store.  `batch_id[i]` is the
 * reference's (event, TPC-group, sub-batch) batch of segment i (cli/simulate_pixels.py:864,902);
 * segments of one batch must be contiguous and batch ids non-decreasing; batch_id < 0 = not simulated. */
int ldsim_segments_upload(ldsim_ctx* ctx, const void* tracks, int64_t n, const LdsimTrackLayout* layout,
                          const int32_t* batch_id);
/* Write the mutated fields (n_electrons, n_photons, pixel_plane, long_diff, tran_diff, t, t_start, t_end)
 * back into host records (D2H). */
int ldsim_segments_download(ldsim_ctx* ctx, void* tracks, int64_t n, const LdsimTrackLayout* layout);
/* Re-unpack the uploaded (device-resident) records into the SoA store, discarding what quench/drift wrote:
 * lets a caller re-run the path on the same resident input without another H2D copy. */
int ldsim_segments_reset(ldsim_ctx* ctx);
/* quench + drift over the resident segments (cli/simulate_pixels.py:732,742) */
int ldsim_dev_quench_drift(ldsim_ctx* ctx, int32_t mode);

typedef struct {
  int64_t n_segments;      /* segments simulated in this call */
  int64_t n_pairs;         /* (segment, pixel) pairs with a valid pixel id */
  int64_t n_unique;        /* unique (batch, pixel) rows produced */
  int64_t n_batches;
  int64_t n_overflow;      /* pixels with more than max_tracks_per_pixel contributing segments */
  int32_t max_active, max_neigh, max_length;
  int32_t n_ambiguous;     /* z slices whose response shift sat within 1e-7 of a rounding boundary */
  int64_t n_dfma;          /* f64 FMAs issued by the induced-current correlation loop (lanes x instructions) */
  int64_t n_fallback;      /* pairs recomputed by the monolithic kernel (split-path capacity overflow) */
  int64_t n_samples;       /* charge samples that passed the bound and were evaluated (2 erf + exp each) */
  int64_t n_wbuf;          /* f64 entries of the weight arena used by the split path in this call */
  int64_t n_dfma_useful;   /* of n_dfma, the FMAs of the quadrature path that are neither padding of an 8-shift weight block
                              nor ticks outside the pair's window: sum over pairs of (weights kept) x (window ticks); 0 when
                              another weights stage ran */
} LdsimChainStats;

/* counters of the last ldsim_tracks_current call (n_segments, n_pairs = S * P, n_fallback, n_wbuf, n_samples, n_dfma,
 * n_dfma_useful, n_ambiguous; the rest 0): which kernels carried the pairs -- n_wbuf > 0 means the split path's weights
 * stage ran, n_fallback counts the pairs the monolithic kernel recomputed */
int ldsim_tracks_current_stats(ldsim_ctx* ctx, LdsimChainStats* stats);

/* Fused a5-a16 (max_pixels .. digitize) on resident segments [seg_begin, seg_end):
 * per unique (batch, pixel), sorted by batch then pixel id exactly like the reference's concatenated
 * per-batch `unique_pix`.  Results stay in HBM; fetch with ldsim_chain_download(). */
int ldsim_charge_chain(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, int32_t want_fractions,
                       LdsimChainStats* stats);
/* Copy the last chain call's rows [0, n_unique) to host.  Any pointer may be NULL.
 *   unique_pix i32[U], batch i32[U], adc_list f64[U][A] (integrated charge), adc_ticks f64[U][A],
 *   adc_digit f64[U][A] (fee.digitize), track_pixel_map i64[U][M] (segment index within the batch, -1 pad),
 *   current_fractions f64[U][A][M] (only if want_fractions). */
int ldsim_chain_download(ldsim_ctx* ctx, int64_t capacity, int32_t* unique_pix, int32_t* batch,
                         double* adc_list, double* adc_ticks, double* adc_digit,
                         int64_t* track_pixel_map, double* current_fractions);
/* The same rows on a second HIP stream, returning at once: the next ldsim_charge_chain computes, into the other of two sets
 * of output buffers, while these rows cross PCIe (the copy engines run beside the kernels).  The host buffers must be
 * page-locked (ldsim_host_alloc) for the copy to be asynchronous, and hold the rows after ldsim_chain_download_wait.  One
 * download is in flight at a time (a second call waits for the first); a launch that would overwrite rows still being
 * copied waits for that copy.  Replaces nothing in the reference (its driver copies with cupy .get() after every batch). */
int ldsim_chain_download_async(ldsim_ctx* ctx, int64_t capacity, int32_t* unique_pix, int32_t* batch, double* adc_list,
                               double* adc_ticks, double* adc_digit, int64_t* track_pixel_map, double* fractions);
int ldsim_chain_download_wait(ldsim_ctx* ctx);
/* The same results in compact form: what the exporter reads (fee.export_to_hdf5, fee.py:143-344: the pixels that hold a hit, their
 * slots up to the first ADC at the pedestal, the fractions of the track slots the pixel has) gathered in HBM, a few MB instead of
 * the dense arrays' 13 KB per unique pixel.  ldsim_chain_compact_build: sizes[4] = hit pixels, hits, track entries (sum over
 * hit pixels of their filled track_pixel_map slots), fraction entries (sum over hits of their pixel's track slots; 0 without
 * want_fractions).  ldsim_chain_compact_download: hit_pixels [n_hp][5] i32 = {row in the dense arrays, pixel id, batch, hits,
 * track slots + 256 if the pixel is the first row of its batch in the dense arrays}, hit pixels in row order; track_segments i64, the filled track_pixel_map entries pixel after pixel; hit_rows
 * [n_hits] 24-byte rows {batch i32, pixel i32, ADC code i32, slot i32, tick f64} and hit_charge f64 (adc_list) in the same
 * order (pixel after pixel, slot 0 up); fractions f64: per hit, one value per track slot of its pixel.  Any pointer may be NULL. */
int ldsim_chain_compact_build(ldsim_ctx* ctx, int64_t* sizes);
int ldsim_chain_compact_download(ldsim_ctx* ctx, int32_t* hit_pixels, int64_t* track_segments, void* hit_rows,
                                 double* hit_charge, double* fractions);
/* Device pointers of the last chain call's compact hit list (for a collective without a host round trip):
 * hits are (batch i32, pixel i32, adc u8-as-i32, tick f64) rows for every written ADC slot. */
int ldsim_chain_compact_hits(ldsim_ctx* ctx, void** dev_rows, int64_t* n_rows, int32_t* row_bytes);

/* ---- device-resident light leg (cli/simulate_pixels.py:749-797 and :1120-1153 on the resident segments) ---------- */
/* lightLUT.calculate_light_incidence[bpg,tpb](tracks, lut, light_sim_dat, track_light_voxel) over ALL resident segments,
 * after ldsim_dev_quench_drift (it needs n_photons and pixel_plane): n_photons_det [n][n_out] f4, t0_det [n][n_out] f4
 * (LIGHT_TRIG_MODE 0 only) and voxel [n][3] i4 stay in HBM; segments outside every TPC hold zeros. */
int ldsim_dev_light_incidence(ldsim_ctx* ctx, int32_t n_out_channels);
/* rows [seg_begin, seg_end) of those arrays to host; any pointer may be NULL */
int ldsim_dev_light_incidence_download(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, float* n_photons_det,
                                       float* t0_det, int32_t* voxel);
/* inputs of light_sim.get_nticks (larndsim/light_sim.py:24-41) for the rows of one batch: min / max of t0_det over the
 * entries with n_photons_det > 0, and whether there is any (trigger mode 0) */
int ldsim_dev_light_t0_range(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, float* t0_min, float* t0_max,
                             int32_t* any);
/* light_sim.sum_light_signals[...] for the resident segments [seg_begin, seg_end) (one batch of the reference's loop):
 * light_sample_inc [n_det][n_ticks] f4 and, with max_truth = MAX_MC_TRUTH_IDS > 0, the truth slots [n_det][n_ticks][max_truth]
 * (i8 ids from segment_track_id[seg_end - seg_begin], f8 photons) are initialised (0 / -1) and filled in HBM.  Segments are
 * visited per detector in descending n_photons_det like the driver's argsort (cli/simulate_pixels.py:1141-1144; equal
 * values, which numpy's unstable sort leaves unpinned, by descending index).
 * Without truth slots (max_truth = 0) the call returns with its kernels in flight on the ctx's stream (every consumer --
 * ldsim_dev_light_download, ldsim_dev_light_response, ldsim_light_kernel_ms -- is ordered after them); the array is brought back to
 * zero by clearing the tick tiles the previous such sum lit when the same buffer serves again (a batch lights its own TPCs' rows
 * only), and op_channel is checked and sent again only when it differs from the last call's. */
int ldsim_dev_sum_light(ldsim_ctx* ctx, int64_t seg_begin, int64_t seg_end, const int32_t* op_channel, int32_t n_det,
                        const int64_t* segment_track_id, int32_t max_truth, double start_time, int32_t n_ticks);
/* results of the last ldsim_dev_sum_light to host; any pointer may be NULL */
int ldsim_dev_light_download(ldsim_ctx* ctx, float* light_sample_inc, int64_t* true_track_id, double* true_photons);
/* HIP-event durations of the last ldsim_dev_light_incidence launch and the last ldsim_dev_sum_light call */
int ldsim_light_kernel_ms(ldsim_ctx* ctx, double* incidence_ms, double* sum_ms);
/* calc_scintillation_effect -> calc_stat_fluctuations (fluctuate != 0; needs ldsim_rng_seed) -> calc_light_detector_response
 * on the photon sum of the last ldsim_dev_sum_light, all in HBM (cli/simulate_pixels.py:1159-1180).  light_gain [n_det] by
 * array row, impulse_model as in ldsim_light_detector_response.  ldsim_light_triggers / ldsim_sim_triggers with a NULL
 * signal then work on the result. */
int ldsim_dev_light_response(ldsim_ctx* ctx, const double* light_gain, const double* impulse_model, int32_t n_impulse,
                             int32_t fluctuate);
/* the three stages' arrays to host ([n_det][n_ticks] f4; truth of the response); any pointer may be NULL */
int ldsim_dev_light_response_download(ldsim_ctx* ctx, float* scint, float* disc, float* response,
                                      int64_t* response_true_track_id, double* response_true_photons);
/* HIP-event durations of the three stages of the last ldsim_dev_light_response */
int ldsim_light_response_ms(ldsim_ctx* ctx, double* scint_ms, double* fluct_ms, double* response_ms);

/* ---- multi-GPU: the one exchange step of the batch-sharded path, RCCL over xGMI -------------------------------
 * One process per GPU; (event, TPC-group) batches are sharded over the ranks and never share a pixel, so the only
 * collective reassembles the output: row counts (all-gather), then the 24-byte hit rows (all-gather-v). */
#define LDSIM_COMM_ID_BYTES 128
/* ncclGetUniqueId on one rank; the host side hands the 128 bytes to every rank (larndsim_amd/comm.py) */
int ldsim_comm_unique_id(void* id);
int ldsim_comm_init(ldsim_ctx* ctx, const void* id, int32_t rank, int32_t world);
int ldsim_comm_destroy(ldsim_ctx* ctx);
/* ranks in the communicator and this rank's index as RCCL reports them (ncclCommCount, ncclCommUserRank); rank may be NULL */
int ldsim_comm_count(ldsim_ctx* ctx, int32_t* n_ranks, int32_t* rank);
/* *value reduced over the ranks (op 0 = sum, 1 = max); also the barrier of the timed region */
int ldsim_comm_allreduce_f64(ldsim_ctx* ctx, double* value, int32_t op);
/* append the last ldsim_charge_chain call's compact hit rows to the pass buffer (reset != 0 empties it first) */
int ldsim_hits_accumulate(ldsim_ctx* ctx, int32_t reset);
/* all-gather-v of the pass buffer: *gathered = device pointer (ctx-owned) to the rows of rank 0, 1, .. back to back,
 * counts[world] (may be NULL) = rows per rank */
int ldsim_comm_allgather_hits(ldsim_ctx* ctx, void** gathered, int64_t* total_rows, int64_t* counts);
int ldsim_comm_gathered_download(ldsim_ctx* ctx, void* rows, int64_t n);

/* ---- fee.export_to_hdf5's hit loop on compact rows (larndsim/fee.py:143-344) -- host code, no ctx, no GPU ---------------------------
 * The LArPix `packets` rows (larpix-control's HDF5 format 2.4: 36-byte packed rows, ldsim_packets_row_bytes) and the
 * `mc_packets_assn` rows (event_ids (1,) i8 | segment_ids (n,) i8 | fraction (n,) f8 | file_traj_ids (n,) i8 | fraction_traj (n,) f8,
 * n = ASSOCIATION_COUNT_TO_STORE: ldsim_packets_assn_row_bytes) of one export, from the rows the reference's loop would enter --
 * the pixels whose first ADC lies above the pedestal code, in export order (batch after batch, pixel after pixel) -- with what is
 * per row precomputed by the caller (larndsim_amd/packets.py: pixel -> io_group / io_channel / chip / channel through tile map,
 * tile orientation and pixel layout, fee.py:150-157,238-255; event start times).  Per hit: clock rollover (:164-183), the packets
 * of a new event (timestamp + sync per io group, the event's light triggers, :187-230), the timestamp packet of a changed tick
 * (:267-277), the data packet with its parity, and the association row (:284-344; equal fractions keep descending slot order --
 * numpy's argsort leaves their order to its build).  Returns the number of rows written, or a negative status. */
typedef struct {
  int64_t n_rows;               /* rows entering the hit loop */
  const int64_t* row_event;     /* [n_rows] event id */
  const int64_t* row_base;      /* [n_rows] int(event_start_time / CLOCK_CYCLE) */
  const int64_t* row_ts_s;      /* [n_rows] int(event_start_time * mus / s): value of the per-event timestamp packets */
  const uint8_t* row_ok;        /* [n_rows] the pixel has a readout address and its channel is not disabled */
  const int32_t* row_io_group;  /* [n_rows] */
  const int32_t* row_io_channel;
  const int32_t* row_chip;
  const int32_t* row_channel;
  const int64_t* row_hit0;      /* [n_rows + 1] offsets into hit_*: the row's slots up to the first ADC <= pedestal */
  const int64_t* row_trk0;      /* [n_rows + 1] offsets into trk_*: the row's filled track slots */
  const int64_t* row_frac0;     /* [n_rows] offset into hit_frac of the row's first hit: [slot][track slot] from there */
  const int32_t* hit_adc;       /* ADC code */
  const double* hit_tick;       /* adc_ticks_list value */
  const double* hit_frac;       /* backtracking fractions */
  const int64_t* trk_segment;   /* segment id of a track slot (track_ids) */
  const int64_t* trk_traj;      /* trajectory id of a track slot (traj_ids) */
  int64_t base0;                /* int(event_start_time / CLOCK_CYCLE) of row 0 of the export's arrays (a row without hits included) */
  int32_t first_row_is_row0;    /* rows[0] is that row 0 */
  int32_t light_trig_mode;      /* light.LIGHT_TRIG_MODE */
  int64_t n_trig;               /* light triggers of the export (fee.py:209-221) */
  const double* trig_time;
  const int64_t* trig_event;
  const int64_t* trig_module;
  int32_t n_io_groups;          /* io groups that get the per-event timestamp / sync packets */
  const int32_t* io_groups;
  int32_t n_modules;            /* MODULE_TO_IO_GROUPS: module ids, and per module its groups module_groups[module_group0[m] .. [m + 1]) */
  const int64_t* module_ids;
  const int32_t* module_group0;
  const int32_t* module_groups;
  int64_t clock_reset_period;   /* detector.CLOCK_RESET_PERIOD */
  double clock_cycle, mus, s;   /* detector.CLOCK_CYCLE, units.mus, units.s */
  int32_t n_keep, max_tracks;   /* sim.ASSOCIATION_COUNT_TO_STORE, sim.MAX_TRACKS_PER_PIXEL */
} LdsimPacketsIn;
int32_t ldsim_packets_row_bytes(void);
int32_t ldsim_packets_assn_row_bytes(int32_t n_keep);
int64_t ldsim_packets_build(const LdsimPacketsIn* in, void* packets_out, void* assn_out, int64_t capacity);

/* ---- output writer helper (host only, no ctx) ----
 * CRC-32 (reflected 0xEDB88320: what a ZIP member carries, = zlib.crc32) of the concatenation of n_parts byte ranges on
 * n_threads host threads (<= 0: all hardware threads).  The driver's .npz writer (cli/simulate_pixels.py OutputFile, the stand-in
 * for the h5py datasets of the reference's cli/simulate_pixels.py:1240-1301) checksums a dataset that exists as the .npy header
 * plus one piece per chain launch without joining the pieces. */
uint32_t ldsim_crc32_parts(const void* const* parts, const uint64_t* sizes, int64_t n_parts, int32_t n_threads);

/* timing of the dominant kernel over the last chain call, measured with HIP events on the ctx stream */
int ldsim_chain_kernel_ms(ldsim_ctx* ctx, double* current_ms, double* adc_ms, double* total_ms);
/* split of current_ms over the tracks_current kernels of the last chain call: weights_kernel, mac_kernel and the
 * monolithic current_kernel (overflow fallback, or the whole stage when the split path is off: then the first two are 0) */
int ldsim_chain_kernel_ms_detail(ldsim_ctx* ctx, double* weights_ms, double* mac_ms, double* fallback_ms);

#ifdef __cplusplus
}
#endif
#endif /* LDSIM_H */
