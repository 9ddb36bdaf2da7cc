// crnn.hip - complex GRU RNN with U(1) mask (J1J2/ComplexRNNwavefunction.py) - placeholder until the
// kernels land: every entry point fails loudly.
#include "models.h"
using namespace rnnwf;
#define NI(h) return (h)->fail(RNNWF_ERR_INVALID, "%s: complex-RNN kernels not built yet", __func__)
int rnnwf::crnn_pack_image(rnnwf_handle* h, std::vector<char>&) { NI(h); }
int rnnwf::crnn_sample(rnnwf_handle* h, int64_t, uint64_t, uint64_t, int64_t, int32_t*, double*) { NI(h); }
int rnnwf::crnn_log_amp(rnnwf_handle* h, const int32_t*, int64_t, float*, double*) { NI(h); }
int rnnwf::crnn_j1j2_eloc(rnnwf_handle* h, const int32_t*, int64_t, const double*, const double*, const double*, int, int, float*, int64_t*) { NI(h); }
int rnnwf::crnn_vmc_step(rnnwf_handle* h, int64_t, uint64_t, uint64_t, int64_t, const double*, int32_t*, float*, double*) { NI(h); }
