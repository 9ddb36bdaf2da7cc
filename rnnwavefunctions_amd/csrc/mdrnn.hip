// mdrnn.hip - 2D MDRNN wave function (2DTFIM_2DRNN/) - placeholder until the kernels land: every entry
// point fails loudly.
#include "models.h"
using namespace rnnwf;
#define NI(h) return (h)->fail(RNNWF_ERR_INVALID, "%s: MDRNN kernels not built yet", __func__)
int rnnwf::mdrnn_pack_image(rnnwf_handle* h, std::vector<char>&) { NI(h); }
int rnnwf::mdrnn_sample(rnnwf_handle* h, int64_t, uint64_t, uint64_t, int64_t, int32_t*, double*) { NI(h); }
int rnnwf::mdrnn_log_prob(rnnwf_handle* h, const int32_t*, int64_t, double*) { NI(h); }
int rnnwf::mdrnn_tfim_eloc(rnnwf_handle* h, const int32_t*, int64_t, const double*, double, double*, double*) { NI(h); }
int rnnwf::mdrnn_vmc_step(rnnwf_handle* h, int64_t, uint64_t, uint64_t, int64_t, const double*, int32_t*, double*, double*) { NI(h); }
