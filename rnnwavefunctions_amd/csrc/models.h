// models.h - host-side entry points of each wave-function family (one .hip translation unit each) and
// the helpers they share (implemented in rnnwf_api.hip).
#pragma once
#include <vector>

#include "handle.h"

namespace rnnwf {

// ---- shared helpers (rnnwf_api.hip) --------------------------------------------------------------
int upload_samples(rnnwf_handle* h, const int32_t* samples, int64_t B);
int pack_device(rnnwf_handle* h, int64_t B, DevBuf& bits, int reverse, const int32_t* col_of_pos_dev);
int unpack_device(rnnwf_handle* h, const DevBuf& bits, int64_t B, const int32_t* pos_of_col_dev);
int upload_and_pack(rnnwf_handle* h, const int32_t* samples, int64_t B, DevBuf& bits, int reverse,
                    const int32_t* col_of_pos_dev);
int unpack_and_download(rnnwf_handle* h, const DevBuf& bits, int64_t B, int32_t* out, const int32_t* pos_of_col_dev);
int run_moments(rnnwf_handle* h, const void* eloc_dev, int64_t ns, bool complex_f32, double* moments_host);
int run_tfim_eloc(rnnwf_handle* h, const uint32_t* bits, const double* lpq, int64_t ns, int Nx, int Ny,
                  const int32_t* pos_of_site_dev, const double* Jz_dev, double Bx, double* eloc_dev);
int run_parity_combine(rnnwf_handle* h, const double* a, const double* b, int64_t n, double* out);
int run_parity_share(rnnwf_handle* h, double* lpF, double* lpR, int64_t n);      // in place: P_F / (P_F + P_R), P_R / (P_F + P_R)

// ---- weight image ---------------------------------------------------------------------------------
int model_pack_image(rnnwf_handle* h, std::vector<char>& img);  // dispatches to the family below
int prnn_pack_image(rnnwf_handle* h, std::vector<char>& img);
int crnn_pack_image(rnnwf_handle* h, std::vector<char>& img);
int mdrnn_pack_image(rnnwf_handle* h, std::vector<char>& img);

// ---- positive GRU RNN (prnn.hip): models GRU1D, GRU1D_PARITY, GRU1D_F64 ---------------------------
int prnn_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, int32_t* out, double* out_log);
int prnn_log_prob(rnnwf_handle* h, const int32_t* samples, int64_t B, double* out);
int prnn_tfim_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, int Nx, int Ny, const double* Jz, double Bx,
                   double* eloc, double* log_probs);
int prnn_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, const double* couplings,
                  int32_t* out_samples, double* out_eloc, double* moments);

// ---- complex GRU RNN with U(1) mask (crnn.hip) -----------------------------------------------------
int crnn_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, int32_t* out, double* out_log);
int crnn_log_amp(rnnwf_handle* h, const int32_t* samples, int64_t B, float* out_re_im, double* out_logp);
int crnn_j1j2_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* J1, const double* J2,
                   const double* Bz, int periodic, int marshall, float* eloc, int64_t* ncon);
int crnn_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, const double* couplings,
                  int32_t* out_samples, float* out_eloc, double* moments);

// ---- 2D MDRNN (mdrnn.hip) ---------------------------------------------------------------------------
int mdrnn_sample(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, int32_t* out, double* out_log);
int mdrnn_log_prob(rnnwf_handle* h, const int32_t* samples, int64_t B, double* out);
int mdrnn_tfim_eloc(rnnwf_handle* h, const int32_t* samples, int64_t ns, const double* Jz, double Bx, double* eloc,
                    double* log_probs);
int mdrnn_vmc_step(rnnwf_handle* h, int64_t ns, uint64_t seed, uint64_t step, int64_t offset, const double* couplings,
                   int32_t* out_samples, double* out_eloc, double* moments);

// ---- bf16x3 engine (split.hip; compiled without SLP packing) -----------------------------------------
struct PrnnArgs;
struct CrnnArgs;
int prnn_split_flip(rnnwf_handle* h, const PrnnArgs& a);
double prnn_split_flops_per_step(rnnwf_handle* h);      // MFMA flops issued per 32-chain wave-step
int prnn_split_pack(rnnwf_handle* h, std::vector<char>& simg);
// split_stream.hip: the same pass at 69..100 units (classic layout, w3 fragments read through L2)
int prnn_split_flip_stream(rnnwf_handle* h, const PrnnArgs& a, int kt16);
double prnn_split_stream_flops_per_step(rnnwf_handle* h);
int prnn_split_stream_pack(rnnwf_handle* h, std::vector<char>& simg);
int prnn_teacher_base(rnnwf_handle* h, int64_t ns, bool reversed, double* out_lp);   // prnn.hip
// the 16x16x32 form at 37..52 units (split_stream.hip; image in h->wsplit16)
int prnn_split_flip_16n(rnnwf_handle* h, const PrnnArgs& a, int kt16);
double prnn_split_16n_flops_per_step();
int prnn_split_16n_pack(rnnwf_handle* h);
int crnn_split_swap_stream(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int kt16);
double crnn_split_stream_flops_per_step(rnnwf_handle* h);
int crnn_split_stream_pack(rnnwf_handle* h, std::vector<char>& simg);
// stacked layers on the bf16x3 engine (split.hip; 37..50 units, ping-pong form): one kernel per layer over the same tiles, the new
// state of every step handed upward through h->xrec (split_core.h: SplitUpperLayout)
bool stack_split_available(const rnnwf_handle* h);
size_t stack_record_bytes_per_32_chains(const rnnwf_handle* h, int64_t steps);   // h->xrec bytes per 32-chain tile column with `steps` wave-steps
int prnn_stack_pack(rnnwf_handle* h);
int prnn_stack_flip(rnnwf_handle* h, const PrnnArgs& a);
int crnn_stack_pack(rnnwf_handle* h);
int crnn_stack_swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles, int64_t max_records);
double stack_split_flops_per_step(rnnwf_handle* h);
int crnn_split_swap(rnnwf_handle* h, const CrnnArgs& a, int64_t max_tiles);
double crnn_split_flops_per_step(rnnwf_handle* h);
int crnn_split_pack(rnnwf_handle* h, std::vector<char>& simg);
// cooperative base pass on the bf16 matrix core (gru_kernels.h: coop_base_pass_bf; f32 models, one layer, num_units <= 52):
// base_bf_available: this handle has the image (h->wbasebf); *_base_coop_bf: the launches; base_bf_pack: at commit
bool base_bf_available(const rnnwf_handle* h);
int base_bf_pack(rnnwf_handle* h);
int prnn_base_coop_bf(rnnwf_handle* h, const PrnnArgs& a);
int crnn_base_coop_bf(rnnwf_handle* h, const CrnnArgs& a);

// teacher-forced base pass with checkpoints on caller-supplied samples (rnnwf_load_batch)
int prnn_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns);
int crnn_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns);
int mdrnn_load_batch(rnnwf_handle* h, const int32_t* samples, int64_t ns);

// ---- gradient (grad.hip) ---------------------------------------------------------------------------
int mdrnn_vmc_gradient(rnnwf_handle* h, double mean_energy, double norm);
// grad_wide.hip: backward kernels compiled in their own translation unit (index: grad.hip, GLaunch::WIDE)
struct GradArgs;
const void* grad_wide_kernel(int which);
void grad_wide_launch(int which, unsigned grid, size_t lds, hipStream_t stream, const GradArgs& a);
int mdrnn_grad_device(rnnwf_handle* h, double mean_energy, double norm, const double* mom_dev, size_t* dw_count);
int mdrnn_pack_table(rnnwf_handle* h, bool backward);
int mdrnn_grad_probe(rnnwf_handle* h, std::vector<int32_t>& sidx, size_t* dw_count);
int grad_single_layer_device(rnnwf_handle* h, double mean_energy, double mean_energy_im, double norm, const double* mom_dev, size_t* dw_count);
int grad_bwd_pack_table(rnnwf_handle* h);
int grad_stack_forward_table(rnnwf_handle* h);
int grad_flat_probe(rnnwf_handle* h, std::vector<int32_t>& sidx, size_t* dw_count, bool* is_f64);
// ---- device-resident training (train.hip) ------------------------------------------------------------
void train_params_changed_on_host(rnnwf_handle* h);           // rnnwf_set_param / rnnwf_commit_params: the device copy is stale
int train_sync_params_to_host(rnnwf_handle* h);               // before the host reads its copy (rnnwf_get_param, checkpoints)
int train_allreduce_grads_device(rnnwf_handle* h);           // 1: gradient all-reduced in-stream on the device, 0: not available
void grad_invalidate(rnnwf_handle* h);
// bytes of per-site hidden states one pass may hold; RNNWF_STATE_BUDGET_MB (read at rnnwf_create) overrides the
// default (tests use it to drive the multi-pass path at small sizes)
size_t state_budget_bytes(const rnnwf_handle* h, size_t dflt);
// RCCL sum of device-resident doubles on the handle's stream (comm.hip); no-op without a communicator
int comm_allreduce_device(rnnwf_handle* h, void* dev, size_t count);
// h->coupl <- n doubles; skipped when they are what the device already holds
int upload_couplings(rnnwf_handle* h, const double* src, size_t n);

}  // namespace rnnwf

extern "C" int rnnwf_comm_destroy(rnnwf_handle* h);
