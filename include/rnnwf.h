/*
 * rnnwf.h - C ABI of librnnwf_hip.so: the MI355X-native VMC inner loop of RNN wave functions.
 *
 * The reference (MatteoMartinelli97/RNNWavefunctions) has no FFI: its boundary for this path is
 * the Python API of its RNNwavefunction classes and local-energy estimators, executed by
 * TensorFlow 1.13 through tf.Session.run.  Each entry point below names the reference interface
 * it stands in for (paths relative to the reference root).  The Python host code under
 * rnnwavefunctions_amd/ binds these symbols with ctypes and re-creates the reference's
 * signatures on top (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C types only; all pointers are HOST pointers owned by the caller unless a name
 *     ends in _dev; the library owns all device memory inside the opaque handle;
 *   - every function returns 0 on success and a negative rnnwf_status otherwise; the message
 *     is available from rnnwf_last_error(); nothing aborts the process;
 *   - a handle is bound to one HIP device and one HIP stream; it is not thread-safe; calls are
 *     synchronous with respect to the host unless stated otherwise;
 *   - spin configurations are int32 arrays of 0/1, row-major (numsamples, N) for the 1D models
 *     and (numsamples, Nx, Ny) for the 2D MDRNN model, exactly as the reference feeds its
 *     placeholders (1DTFIM/TrainingRNN_1DTFIM.py:192).
 */
#ifndef RNNWF_H
#define RNNWF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNNWF_ABI_VERSION 1

typedef struct rnnwf_handle rnnwf_handle;

typedef enum {
    RNNWF_OK = 0,
    RNNWF_ERR_INVALID = -1,     /* bad argument / unsupported configuration            */
    RNNWF_ERR_HIP = -2,         /* a HIP runtime call failed                           */
    RNNWF_ERR_STATE = -3,       /* call order violated (e.g. parameters not committed) */
    RNNWF_ERR_COMM = -4,        /* RCCL failure                                        */
    RNNWF_ERR_NOMEM = -5
} rnnwf_status;

/* Which reference wave function the handle implements. */
typedef enum {
    RNNWF_MODEL_GRU1D = 0,       /* 1DTFIM/RNNwavefunction.py:7-118          pRNN, f32 cell, f64 log-sum   */
    RNNWF_MODEL_GRU1D_PARITY = 1,/* 1DTFIM/RNNwavefunction_paritysym.py:7-145 same sampler, symmetrised P   */
    RNNWF_MODEL_CRNN_U1 = 2,     /* J1J2/ComplexRNNwavefunction.py:15-169    cRNN, f32/complex64, U(1) mask */
    RNNWF_MODEL_GRU1D_F64 = 3,   /* 2DTFIM_1DRNN/RNNwavefunction.py:8-130    pRNN over a raster path, f64   */
    RNNWF_MODEL_MDRNN2D = 4      /* 2DTFIM_2DRNN/RNNwavefunction.py:5-200 + MDRNNcell.py:6-66, f64          */
} rnnwf_model;

typedef enum { RNNWF_F32 = 0, RNNWF_F64 = 1 } rnnwf_dtype;

#define RNNWF_MAX_LAYERS 4

typedef struct {
    int32_t abi_version;              /* RNNWF_ABI_VERSION                                        */
    int32_t model;                    /* rnnwf_model                                              */
    int32_t nx;                       /* systemsize (1D) or systemsize_x (2D)                     */
    int32_t ny;                       /* 1 for 1D models, systemsize_y for 2D                     */
    int32_t num_layers;               /* len(units) of the reference ctor: 1; 2..4 (any widths     */
                                      /* <= 100 units, float64: <= 68) for the GRU models          */
    int32_t units[RNNWF_MAX_LAYERS];  /* units[n]: one layer <= 260 (float GRU models; above 100 the */
                                      /* weight image is read through L2), <= 100 (float64 GRU;   */
                                      /* above 68 likewise), <= 84 (2D RNN)                       */
    int32_t device;                   /* HIP device ordinal                                       */
    int32_t reserved[6];
} rnnwf_config;

/* ---- life cycle ------------------------------------------------------------------------------
 * rnnwf_create      <- RNNwavefunction.__init__ (1DTFIM/RNNwavefunction.py:8-33,
 *                      J1J2/ComplexRNNwavefunction.py:16-43, 2DTFIM_2DRNN/RNNwavefunction.py:6-33)
 *                      plus the tf.Session it would be run in (1DTFIM/TrainingRNN_1DTFIM.py:119-123). */
int rnnwf_create(const rnnwf_config* cfg, rnnwf_handle** out);
int rnnwf_destroy(rnnwf_handle* h);
/* Message of the last failure on this handle (h == NULL: of the last failed rnnwf_create). */
const char* rnnwf_last_error(const rnnwf_handle* h);
/* "hip-gfx950" */
const char* rnnwf_backend_name(void);
int rnnwf_abi_version(void);

/* ---- parameters ------------------------------------------------------------------------------
 * Stand in for the TF variables of the reference graph and for tf.train.Saver restore
 * (1DTFIM/TrainingRNN_1DTFIM.py:166,172-183).  `tf_name` is the TF variable name WITHOUT the scope
 * prefix, e.g. "multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/gates/kernel", "wf_dense/bias",
 * "Wh_rnn_0".  `count` is the element count and must match the variable's shape; `dtype` is the
 * type of the caller's buffer (converted to the model's arithmetic type).
 * rnnwf_commit_params re-packs all parameters into the MFMA-fragment image the kernels stage in
 * LDS and uploads it; it must be called after the last rnnwf_set_param and before any compute. */
int rnnwf_set_param(rnnwf_handle* h, const char* tf_name, const void* data, int64_t count, int32_t dtype);
int rnnwf_get_param(rnnwf_handle* h, const char* tf_name, void* data, int64_t count, int32_t dtype);
int rnnwf_commit_params(rnnwf_handle* h);
/* rnnwf_init_params <- sess.run(tf.global_variables_initializer()) (1DTFIM/TrainingRNN_1DTFIM.py:168): glorot/xavier-
 * uniform kernels, gate bias 1, other biases 0 (MDRNN: all tensors xavier, MDRNNcell.py:21-35), drawn in the order
 * and with the generator of the Python side (numpy.random.RandomState(seed): MT19937, 53-bit doubles), so C and
 * Python callers start from identical weights; commits.  TensorFlow's own seeded draws are not reproducible outside
 * TF ("parity unpinned").  seed must fit 32 bits. */
int rnnwf_init_params(rnnwf_handle* h, uint64_t seed);
/* Number of scalar parameters (the count the reference prints, TrainingRNN_1DTFIM.py:127-136). */
int64_t rnnwf_num_params(const rnnwf_handle* h);

/* ---- wave function ---------------------------------------------------------------------------
 * rnnwf_sample   <- sess.run(wf.sample(numsamples, 2))   (1DTFIM/RNNwavefunction.py:35-74,
 *                   J1J2/ComplexRNNwavefunction.py:45-103, 2DTFIM_2DRNN/RNNwavefunction.py:35-118).
 *   Draws `numsamples` configurations.  The uniform of (global sample g = sample_offset + row,
 *   site n) is Philox4x32-10(key = seed, counter = (g, n/4, step))[n%4] >> 8, so shards of one
 *   batch drawn with different sample_offset on different devices are disjoint pieces of the
 *   same stream.  out_samples: (numsamples, N) int32; out_log (optional, may be NULL): the
 *   log-probability (f64) of each drawn configuration (2*Re log psi for the cRNN).            */
int rnnwf_sample(rnnwf_handle* h, int64_t numsamples, uint64_t seed, uint64_t step, int64_t sample_offset,
                 int32_t* out_samples, double* out_log);

/* rnnwf_log_prob <- sess.run(wf.log_probability(ph, 2), {ph: samples})
 *                   (1DTFIM/RNNwavefunction.py:76-118, RNNwavefunction_paritysym.py:80-145,
 *                    2DTFIM_1DRNN/RNNwavefunction.py:84-130, 2DTFIM_2DRNN/RNNwavefunction.py:120-200)
 *   out: (B,) f64.  For RNNWF_MODEL_CRNN_U1 it returns 2*Re log psi.                          */
int rnnwf_log_prob(rnnwf_handle* h, const int32_t* samples, int64_t B, double* out);

/* rnnwf_log_amp  <- sess.run(wf.log_amplitude(ph, 2), {ph: samples})
 *                   (J1J2/ComplexRNNwavefunction.py:105-169).  out_re_im: (B, 2) f32 = complex64. */
int rnnwf_log_amp(rnnwf_handle* h, const int32_t* samples, int64_t B, float* out_re_im);

/* ---- local-energy estimators -----------------------------------------------------------------
 * rnnwf_tfim_eloc <- Ising_local_energies(Jz, Bx, samples, queue_samples, log_probs_tensor,
 *                    samples_placeholder, log_probs, sess)   (1DTFIM/TrainingRNN_1DTFIM.py:13-75)
 *   Fused formulation: one teacher-forced pass that checkpoints the hidden state after every
 *   site, then every single-spin-flip configuration is scored from its flipped site onwards only;
 *   `queue_samples` is never materialised.  Jz: (N,) f64 (Jz[N-1] unused, as in the reference).
 *   eloc: (numsamples,) f64.  log_probs (optional): ((N+1)*numsamples,) f64, row 0 = log P(s),
 *   row i+1 = log P(s with spin i flipped) - the reference's `log_probs` scratch (:65,:70).   */
int rnnwf_tfim_eloc(rnnwf_handle* h, const int32_t* samples, int64_t numsamples, const double* Jz, double Bx,
                    double* eloc, double* log_probs);

/* rnnwf_tfim2d_eloc <- Ising2D_local_energies(Jz, Bx, Nx, Ny, samples, ...)
 *   (2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:13-83 for RNNWF_MODEL_MDRNN2D, samples (ns, Nx, Ny);
 *    2DTFIM_1DRNN/Training1DRNN_2DTFIM.py:13-81 for RNNWF_MODEL_GRU1D_F64, samples (ns, Nx*Ny)).
 *   Jz: (Nx, Ny) f64 row-major.                                                                */
int rnnwf_tfim2d_eloc(rnnwf_handle* h, const int32_t* samples, int64_t numsamples, const double* Jz, double Bx,
                      double* eloc, double* log_probs);

/* rnnwf_j1j2_eloc <- J1J2Slices + chunked log_amplitude + E_loc loop
 *   (J1J2/TrainingRNN_J1J2.py:95-127, :255-279), with J1J2MatrixElements' `periodic` and
 *   `Marshall_sign` flags (:12) implemented as documented (the reference's J1J2Slices passes
 *   Marshall_sign into the `periodic` slot, :118; the Python facade reproduces that quirk).
 *   J1, J2, Bz: (N,) f64.  eloc_re_im: (numsamples, 2) f32.  n_connected (optional): total
 *   number of configurations scored (diagonal + off-diagonal), the reference's `len_sigmas`.  */
int rnnwf_j1j2_eloc(rnnwf_handle* h, const int32_t* samples, int64_t numsamples, const double* J1,
                    const double* J2, const double* Bz, int32_t periodic, int32_t marshall,
                    float* eloc_re_im, int64_t* n_connected);

/* ---- fused VMC step (sample + local energy + moments, nothing leaves HBM but the moments) -----
 * rnnwf_vmc_step <- one iteration of the loop at 1DTFIM/TrainingRNN_1DTFIM.py:199-207 /
 *   J1J2/TrainingRNN_J1J2.py:241-282 / the Training scripts of the two 2DTFIM folders, without the optimizer:
 *   samples = sess.run(samples_); local_energies = <estimator>(...); meanE; varE.
 *   `couplings` is model specific: TFIM1D: Jz (N) then Bx (1)              -> N+1 doubles
 *                                  TFIM2D: Jz (Nx*Ny) then Bx (1)          -> Nx*Ny+1 doubles
 *                                  J1J2  : J1 (N), J2 (N), Bz (N), periodic, marshall -> 3N+2 doubles
 *   moments[4] (out) = { sum Re E, sum (Re E)^2, n, sum Im E } over THIS handle's samples
 *   (combine across devices with rnnwf_allreduce_moments).  out_samples / out_eloc may be NULL
 *   (out_eloc is (numsamples,) f64 for TFIM, (numsamples,2) f32 for J1J2).                     */
int rnnwf_vmc_step(rnnwf_handle* h, int64_t numsamples, uint64_t seed, uint64_t step, int64_t sample_offset,
                   const double* couplings, int64_t n_couplings, int32_t* out_samples, void* out_eloc,
                   double* moments);

/* ---- gradient of the VMC cost (models GRU1D, GRU1D_PARITY, CRNN_U1 (f32), GRU1D_F64, MDRNN2D (f64)) ----
 * rnnwf_vmc_gradient <- optimizer.compute_gradients(cost) with
 *   cost = mean(log_probs * Eloc) - mean(Eloc) * mean(log_probs)      (1DTFIM/TrainingRNN_1DTFIM.py:151-162)
 *   evaluated on the batch of the LAST rnnwf_vmc_step (its samples, per-site hidden states and E_loc are still
 *   resident):  grad = sum_s (E_s - mean_energy) / norm * d log P(s) / d theta.  Single device: mean_energy =
 *   moments[0]/moments[2], norm = numsamples; sharded: the all-reduced mean and the global sample count, then
 *   rnnwf_allreduce_grads.  Back-propagation through time on the MFMA + a TN GEMM for the weight gradients.
 *   Complex RNN: cost = 2 Re(mean(conj(log_amplitudes) Eloc) - conj(mean(log_amplitudes)) mean(Eloc))
 *   (J1J2/TrainingRNN_J1J2.py:197), i.e. grad = 2/norm sum_s [(Re E_s - mean_energy) d Re log psi +
 *   (Im E_s - mean_energy_im) d Im log psi]; mean_energy_im is ignored for the positive RNNs.
 *   The 2D drivers (2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:163, 2DTFIM_1DRNN/Training1DRNN_2DTFIM.py:160) use the
 *   first cost in float64.  Every width rnnwf_create accepts (above 68 / 52 units the backward operand is read through
 *   L2 instead of LDS).  Stacked layers (len(units) 2..RNNWF_MAX_LAYERS, every GRU model): one backward pass per
 *   layer, top first.  GRU1D_PARITY (the import switch of 1DTFIM/TrainingRNN_1DTFIM.py:10): log P_sym =
 *   log(0.5 (P(s) + P(reversed s))), two backward passes weighted by each direction's share of P_sym.
 *   Every reduction has a fixed order: the same batch gives the same bits.
 * rnnwf_get_grad     <- the gradient of one TF variable (same names and shapes as rnnwf_set_param).
 * rnnwf_allreduce_grads: one RCCL all-reduce (sum) over all gradient arrays of the handle (flattened and summed on the
 *   device, in-stream, then copied to the host arrays; through pinned staging when the device-side map is unavailable).  */
int rnnwf_vmc_gradient(rnnwf_handle* h, double mean_energy, double mean_energy_im, double norm);
/* rnnwf_load_batch: the batch rnnwf_vmc_gradient works on, supplied by the caller instead of drawn by rnnwf_vmc_step -
 * what `sess.run(optstep, feed_dict={Eloc: local_energies, samp: samples, ...})` feeds (TrainingRNN_1DTFIM.py:221,
 * TrainingRNN_J1J2.py:286).  samples: int32 (ns, N) as for rnnwf_log_prob; eloc: float64[ns] (TFIM models) or
 * complex64[ns] as float pairs (complex RNN).  Runs the teacher-forced pass that stores the hidden-state checkpoints.  */
int rnnwf_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns, const void* eloc);
int rnnwf_get_grad(rnnwf_handle* h, const char* tf_name, void* data, int64_t count, int32_t dtype);
/* One call per training iteration instead of one per tensor: every tensor in the reference's shapes, float64, concatenated in
 * the byte-wise order of the names (rnnwf_param_name(h, i, &count) lists them).  rnnwf_set_params_flat commits. */
int rnnwf_set_params_flat(rnnwf_handle* h, const double* flat, int64_t count);
int rnnwf_get_grads_flat(rnnwf_handle* h, double* flat, int64_t count);
const char* rnnwf_param_name(const rnnwf_handle* h, int32_t i, int64_t* count);
int rnnwf_allreduce_grads(rnnwf_handle* h);

/* ---- device-resident training iteration (every model of the four drivers) ----------------------------------------------
 * The reference's update is one `sess.run(optstep)` on the device (1DTFIM/TrainingRNN_1DTFIM.py:113,162,221;
 * J1J2/TrainingRNN_J1J2.py:164,286): tf.train.AdamOptimizer on the gradient of the cost, variables never leave the device.
 * rnnwf_device_training_supported: 1 when this handle can do the same (parameters, Adam moments and gradient resident on the
 *   device, the kernels' weight images rebuilt there after every update) - every GRU model, one layer or a stack, and the 2D
 *   RNN; 0 (a width without a packer table): keep the host optimizer (rnnwf_get_grads_flat / rnnwf_set_params_flat).
 * rnnwf_adam_step  <- optimizer.apply_gradients on the gradient the last rnnwf_vmc_gradient left on the device:
 *   t += 1; lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t); m = beta1 m + (1 - beta1) g; v = beta2 v + (1 - beta2) g^2;
 *   theta -= lr_t m / (sqrt(v) + epsilon) in float64, theta rounded to the model's type; then every weight image is rebuilt on the device.
 * rnnwf_train_steps <- K iterations of the loop at TrainingRNN_1DTFIM.py:199-227 (sample, local energies, moments, gradient,
 *   update) with ONE host synchronisation: iteration k draws with step index step0 + k and uses learning_rates[k];
 *   moments: [K][4] as rnnwf_vmc_step returns them (summed over the ranks when rnnwf_comm_reduce_in_step is on, as is the
 *   gradient: one in-stream RCCL all-reduce each).  1 <= K <= 1024.  Same arithmetic as the host optimizer, operation for
 *   operation: the trajectory is bit-identical.
 * rnnwf_adam_get_state / rnnwf_adam_set_state: Adam's m and v in the flat order of rnnwf_set_params_flat and the number of
 *   updates applied (what tf.train.Saver keeps as <var>/Adam, <var>/Adam_1 and the beta powers); NULL m/v in set: zeros.
 * rnnwf_get_param and rnnwf_commit_params see the device's parameters (they are copied back on demand).                    */
int rnnwf_device_training_supported(rnnwf_handle* h);
int rnnwf_adam_step(rnnwf_handle* h, double learning_rate, double beta1, double beta2, double epsilon);
int rnnwf_train_steps(rnnwf_handle* h, int32_t K, int64_t numsamples, uint64_t seed, uint64_t step0, int64_t sample_offset,
                      const double* couplings, int64_t n_couplings, const double* learning_rates, double beta1, double beta2,
                      double epsilon, double* moments);
int rnnwf_adam_get_state(rnnwf_handle* h, double* m_flat, double* v_flat, int64_t count, int64_t* t_steps);
int rnnwf_adam_set_state(rnnwf_handle* h, const double* m_flat, const double* v_flat, int64_t count, int64_t t_steps);

/* ---- multi-GPU: one RCCL all-reduce of the energy moments -------------------------------------
 * The reference is single-process; these add the one data-parallel collective of SURVEY.md 8e.
 * One process per GPU: rank 0 calls rnnwf_comm_unique_id and ships the 128 bytes to the other
 * ranks by any means (the Python host uses the launcher's store); every rank then calls
 * rnnwf_comm_init.  rnnwf_allreduce_moments sums `count` doubles in place over all ranks
 * (ncclAllReduce, ncclDouble, ncclSum on the handle's stream) - population mean/variance follow
 * as S1/n and S2/n - (S1/n)^2 (np.mean / np.var, TrainingRNN_1DTFIM.py:206-207).              */
#define RNNWF_UNIQUE_ID_BYTES 128
int rnnwf_comm_unique_id(void* id_out);
int rnnwf_comm_init(rnnwf_handle* h, const void* id, int32_t rank, int32_t nranks);
int rnnwf_allreduce_moments(rnnwf_handle* h, double* moments, int32_t count);
/* The same all-reduce for `count` doubles of any length (gradient partial sums held by the caller, histories ...):
 * host array -> pinned staging -> device -> ncclAllReduce(sum) on the handle's stream -> back; identity on one rank. */
int rnnwf_allreduce_f64(rnnwf_handle* h, double* data, int64_t count);
/* on != 0: rnnwf_vmc_step itself returns the moments summed over all ranks - the all-reduce runs on the handle's stream on
 * the device-resident moments, in front of the step's single host synchronisation (no second round trip per step).   */
int rnnwf_comm_reduce_in_step(rnnwf_handle* h, int32_t on);
/* The communicator's own answer (ncclCommCount, ncclCommUserRank) and the handle's device ordinal; 1 / 0 when no
 * communicator exists.  Reports print it, so that N one-rank runs cannot be mistaken for one N-rank run.    */
int rnnwf_comm_info(rnnwf_handle* h, int32_t* nranks, int32_t* rank, int32_t* device);
int rnnwf_comm_destroy(rnnwf_handle* h);

/* ---- measurement ------------------------------------------------------------------------------
 * HIP-event timing of the kernels on the handle's stream (bench.py's roofline leg).
 * kernel ids: 0 = base pass (sample / teacher-forced + checkpoints), 1 = flip pass (dominant),
 *             2 = local-energy assembly + moments, 3 = back-propagation through time of rnnwf_vmc_gradient,
 *             4 = its weight-gradient GEMM.  total_ms / launches accumulate since the
 *             last rnnwf_timing_reset.  Stacked layers on the bf16x3 engine: id 1 brackets the whole
 *             pipeline of per-layer kernels as ONE launch.  work[] is the same for every id: work[0] =
 *             cell evaluations (all layers of a chain step count as one), work[1] = MFMA flops issued
 *             (padding included), both by the flip / swap pass (id 1) since the last reset.
 * rnnwf_timing_enable: on = 0 off, 1 all groups, 2 the dominant pass (id 1) only - two events per step instead of
 *             ten, which is what a throughput measurement wants beside its roofline figure.       */
int rnnwf_timing_enable(rnnwf_handle* h, int32_t on);
int rnnwf_timing_reset(rnnwf_handle* h);
int rnnwf_timing_get(rnnwf_handle* h, int32_t kernel_id, double* total_ms, int64_t* launches, double* work);
/* Which matrix engine the dominant (flip / swap) pass of this handle uses, decided at rnnwf_commit_params:
 *   "bf16x3"  - both operands held exactly as three bf16 parts, six bf16 MFMA products, f32 accumulate
 *               (f32 models up to 100 units; f32 accuracy, see csrc/split_core.h; above 68 units one weight part is read
 *               through L2; stacked layers of 37..50 units: one kernel per layer, csrc/split_kernels.h).  With one layer of
 *               37..52 units the base pass (sampling, log_probability) runs on the same engine (cooperative kernel);
 *   "f32mfma" - f32-input MFMA: forced with RNNWF_ENGINE=f32; above 100 units; stacked layers of other widths; batches too
 *               small to fill the chip with 32-chain tiles; every other base pass;
 *   "f64mfma" - the float64 models.                                                                     */
const char* rnnwf_engine_name(const rnnwf_handle* h);
/* hipDeviceSynchronize on the handle's device (bench.py brackets its timed region with it). */
int rnnwf_synchronize(rnnwf_handle* h);
/* Device properties for reports: cu_count, clock_mhz, hbm_bytes. */
int rnnwf_device_info(rnnwf_handle* h, int32_t* cu_count, int32_t* clock_mhz, int64_t* hbm_bytes, char* name64);

#ifdef __cplusplus
}
#endif
#endif /* RNNWF_H */
