// pack_value.h - the two scalar types the host-side image packers (pack.h, pack_split.h, grad.hip) are written over.
//
// S = double : the packer fills the image with the values of the committed parameters (what rnnwf_commit_params uploads).
// S = Lin    : the SAME code, run once per handle, records for every image element WHICH parameters it is made of and how:
//                  value = c * ((P[a] + s * P[b]) + P[d])   (b, d optional; a bias plus an input row, a head column difference, the 2D
//                                                            cell's b + Uh[x_h] + Uv[x_v], ...)
//              followed by the store's conversion (f32, f64, or part k of the exact three-way bf16 split of the f32 value).
//              The record - a PackTable - is what the device-side re-pack kernel (train.hip: repack_kernel) replays on the
//              device-resident parameters after every optimizer step, so that a training iteration never visits the host
//              (the reference runs `sess.run(optstep)` on the device, 1DTFIM/TrainingRNN_1DTFIM.py:113,162,221).
//              Because it is the packer's own arithmetic that is recorded (same operation order: t = P[a] +- P[b] in double, + P[d],
//              then c * t, then the conversion), the replayed image equals the host-packed one bit for bit (tests/test_gpu_training.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "handle.h"

namespace rnnwf {

inline uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_float(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// w -> three bf16 numbers whose sum is exactly (float)w
inline void split3(double w, uint16_t (&p)[3]) {
    float r = (float)w;
    for (int i = 0; i < 3; ++i) {
        p[i] = bf16_rne(r);
        r -= bf16_to_float(p[i]);
    }
}

// c * ((P[a] + s * P[b]) + P[d]); a < 0: the constant v.  v always holds the value for the parameters the packer sees.
struct Lin {
    double v = 0.0;
    int32_t a = -1, b = -1, d = -1;
    double c = 1.0;
    int32_t s = 0;
    Lin() = default;
    Lin(double k) : v(k) {}                                   // a constant (0.0 for padded units)
};
inline Lin operator+(const Lin& x, const Lin& y) {
    if (x.a < 0 && x.v == 0.0) return y;
    if (y.a < 0 && y.v == 0.0) return x;
    if (x.a < 0 || y.a < 0 || x.d >= 0 || y.b >= 0 || y.d >= 0 || x.c != 1.0 || y.c != 1.0) throw std::logic_error("pack_value.h: unsupported sum in a packer");
    Lin r = x;
    r.v = x.v + y.v;
    if (x.b < 0) { r.b = y.a; r.s = 1; }
    else r.d = y.a;                                           // (P[a] +- P[b]) + P[d]
    return r;
}
inline Lin operator-(const Lin& x, const Lin& y) {
    if (y.a < 0 && y.v == 0.0) return x;
    if (x.a < 0 || y.a < 0 || x.b >= 0 || y.b >= 0 || x.d >= 0 || y.d >= 0 || x.c != 1.0 || y.c != 1.0) throw std::logic_error("pack_value.h: unsupported difference in a packer");
    Lin r;
    r.v = x.v - y.v; r.a = x.a; r.b = y.a; r.s = -1;
    return r;
}
inline Lin operator*(double k, const Lin& x) {
    if (x.c != 1.0) throw std::logic_error("pack_value.h: a packer scales a value twice");
    Lin r = x;
    r.v = k * x.v;
    r.c = k;
    return r;
}

// one element of an image as the device re-pack kernel rebuilds it
struct PackEntry {
    uint32_t off;        // byte offset in the image
    int32_t a, b;        // flat parameter indices (order of rnnwf_set_params_flat), b < 0: none
    int32_t kind;        // 0: f32, 1: f64, 2 + k: bf16 part k of the f32 value; bit 8: s = -1 (difference instead of sum)
    int32_t d;           // third term (added after a +- b), < 0: none
    double c;            // scale (1.0: none)
};
struct PackTable {
    std::vector<PackEntry> entries;
    size_t image_bytes = 0;
};
enum { PACK_F32 = 0, PACK_F64 = 1, PACK_BF16 = 2, PACK_MINUS = 256 };

// where Lin stores go while a packer runs in recording mode (one packer at a time per thread)
struct PackTrace {
    const char* base = nullptr;
    PackTable* table = nullptr;
    size_t shift = 0;            // where the image being packed starts inside the device buffer the table rebuilds (stacked layers: one
                                 // buffer holds the images of all layers, one after the other)
};
inline PackTrace& pack_trace() {
    static thread_local PackTrace t;
    return t;
}

template <class S> struct PackSink;
template <> struct PackSink<double> {
    static void begin(const std::vector<char>&) {}
    template <class T> static void put(T* dst, double x) { *dst = (T)x; }
    static void put_parts(uint16_t* d0, uint16_t* d1, uint16_t* d2, double x) {
        uint16_t p[3];
        split3(x, p);
        *d0 = p[0]; *d1 = p[1]; *d2 = p[2];
    }
    static void put_part(uint16_t* d, double x, int k) {
        uint16_t p[3];
        split3(x, p);
        *d = p[k];
    }
};
template <> struct PackSink<Lin> {
    static void begin(const std::vector<char>& img) {
        pack_trace().base = img.data();
        if (pack_trace().table) pack_trace().table->image_bytes = std::max(pack_trace().table->image_bytes, pack_trace().shift + img.size());
    }
    static void record(const void* dst, const Lin& x, int kind) {
        PackTrace& t = pack_trace();
        if (!t.table || x.a < 0) return;                      // constants stay what the host-packed image holds
        PackEntry e;
        e.off = (uint32_t)((const char*)dst - t.base + t.shift);
        e.a = x.a; e.b = x.b; e.d = x.d;
        e.kind = kind | (x.b >= 0 && x.s < 0 ? PACK_MINUS : 0);
        e.c = x.c;
        t.table->entries.push_back(e);
    }
    static void put(float* dst, const Lin& x) { record(dst, x, PACK_F32); }
    static void put(double* dst, const Lin& x) { record(dst, x, PACK_F64); }
    static void put_parts(uint16_t* d0, uint16_t* d1, uint16_t* d2, const Lin& x) {
        record(d0, x, PACK_BF16 + 0); record(d1, x, PACK_BF16 + 1); record(d2, x, PACK_BF16 + 2);
    }
    static void put_part(uint16_t* d, const Lin& x, int k) { record(d, x, PACK_BF16 + k); }
};

// parameter tensors as the packers index them (padded to the widest layer, handle.h: ParamSpec)
template <class S> struct ParamView;
template <> struct ParamView<double> {
    const std::vector<double>* v;
    double operator[](size_t i) const { return (*v)[i]; }
};
template <> struct ParamView<Lin> {
    const std::vector<double>* v;
    const std::vector<int32_t>* flat;                         // padded index -> flat parameter index, -1: padding (always 0)
    Lin operator[](size_t i) const {
        Lin x;
        x.v = (*v)[i];
        x.a = (*flat)[i];
        if (x.a < 0) x.v = 0.0;
        return x;
    }
};

}  // namespace rnnwf
