// init_params.h - host-only: the initial parameter values of rnnwavefunctions_amd/params.py, bit for bit.
//
// params.py draws glorot/xavier-uniform kernels from numpy.random.RandomState(seed) in a fixed order (gate bias 1,
// other biases 0; MDRNN: all five tensors xavier, incl. b).  numpy's RandomState(int seed) is MT19937 seeded with
// init_genrand(seed), a double is genrand_res53 ((a >> 5) * 2^26 + (b >> 6)) / 2^53, and
// uniform(low, high) = low + (high - low) * double - reproduced here so that a C caller of rnnwf_init_params gets the
// values a Python caller gets from init_gru_params / init_mdrnn_params.  (TensorFlow's own seeded initialisers are
// not reproducible outside TF: SURVEY.md 8c "parity unpinned".)
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace rnnwf {

class NumpyRandomState {
public:
    explicit NumpyRandomState(uint32_t seed) {
        mt_[0] = seed;
        for (int i = 1; i < 624; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
        pos_ = 624;
    }
    uint32_t next_u32() {
        if (pos_ >= 624) refill();
        uint32_t y = mt_[pos_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double next_double() {
        const uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    double uniform(double low, double high) { return low + (high - low) * next_double(); }

private:
    void refill() {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt_[k] & 0x80000000u) | (mt_[(k + 1) % 624] & 0x7fffffffu);
            mt_[k] = mt_[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos_ = 0;
    }
    uint32_t mt_[624];
    int pos_;
};

// _glorot(rng, shape, dtype) of params.py: rows x cols values in C order, rounded to `f32` if asked
inline void glorot_fill(NumpyRandomState& rng, int64_t rows, int64_t cols, bool vector, bool f32, std::vector<double>& out) {
    const double fan_in = vector ? (double)cols : (double)rows, fan_out = (double)cols;
    const double limit = std::sqrt(6.0 / (fan_in + fan_out));
    out.resize((size_t)(rows * cols));
    for (auto& v : out) {
        const double x = rng.uniform(-limit, limit);
        v = f32 ? (double)(float)x : x;
    }
}

}  // namespace rnnwf
