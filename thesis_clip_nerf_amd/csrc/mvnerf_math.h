// Scalar fp32 building blocks of the MVNeRF render path, usable from device code (hipcc) and from
// a host build (gcc, tests/test_device_math_cpu.py) so the exact same source is checked on CPU.
//
// Arithmetic contract (mirrors oracle/mvnerf_oracle.py): one IEEE rounding per written operation,
// no fused multiply-add on the geometry chain.  The whole library is compiled with
// -ffp-contract=off, so `a * b + c` below is a rounded multiply followed by a rounded add; fmaf()
// is used only where an FMA is wanted (inside sin/cos, which are compared with a tolerance).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MV_HD __host__ __device__ __forceinline__
#else
#define MV_HD static inline
#endif

namespace mvnerf {

constexpr int kNFreq = 10;
constexpr int kHidden = 128;
constexpr int kFeat = 256;
constexpr int kIn = 379;          // 60 + 60 + 3 + 256 (layers.py:361)
constexpr int kBlocks = 6;
constexpr int kNetParams = 247300;

// ---- Keras-order flat buffer offsets (floats) --------------------------------------------
constexpr int kKerasW0 = 0;
constexpr int kKerasB0 = kIn * kHidden;                         // 48512
constexpr int kKerasBlocks = kKerasB0 + kHidden;                // 48640
constexpr int kKerasBlockStride = 2 * (kHidden * kHidden + kHidden);   // 33024
constexpr int kKerasWr = kKerasBlocks + kBlocks * kKerasBlockStride;   // 246784
constexpr int kKerasBr = kKerasWr + kHidden * 4;                // 247296

// ---- a8: sin/cos of an fp32 argument, |x| up to ~1e5, ~1.5 ulp ---------------------------------
// Three-term Cody-Waite reduction by pi/2 with FMAs, then minimax polynomials on [-pi/4, pi/4].
// position_encoding (nerf_utils.py:120-123) forms the fp32 product x*fl32(pi*2^k) first and takes
// sin/cos of that rounded value; callers pass exactly that product.
MV_HD void sincos_reduced(float r, int q, float* s_out, float* c_out) {
    const float s2 = r * r;
    float ps = 2.86567956e-6f;
    ps = fmaf(ps, s2, -1.98559923e-4f);
    ps = fmaf(ps, s2, 8.33338592e-3f);
    ps = fmaf(ps, s2, -1.66666672e-1f);
    const float t = r * s2;
    const float sn = fmaf(ps, t, r);
    float pc = 2.44677067e-5f;
    pc = fmaf(pc, s2, -1.38877297e-3f);
    pc = fmaf(pc, s2, 4.16666567e-2f);
    pc = fmaf(pc, s2, -5.00000000e-1f);
    const float cs = fmaf(pc, s2, 1.0f);
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}

// |x| >= 1e5 (camera-space coordinates beyond ~60 m at the top octave) and non-finite x: two-term
// Cody-Waite in float64.  Accurate to ~1e-7 up to |x| ~ 1e9; beyond that the fp32 argument spacing
// exceeds 2*pi and the value is numerically meaningless in the reference as well.
MV_HD void sincos_large(float x, float* s_out, float* c_out) {
    const double xd = (double)x;
    const double j = rint(xd * 0.63661977236758134308);
    double r = fma(j, -1.57079632679489655800e+00, xd);
    r = fma(j, -6.12323399573676603587e-17, r);
    const double jq = j - 4.0 * floor(j * 0.25);          // j mod 4, exact while |j| < 2^53
    sincos_reduced((float)r, (int)jq, s_out, c_out);
}

MV_HD void sincos_f32(float x, float* s_out, float* c_out) {
    if (!(fabsf(x) < 100000.0f)) {
        sincos_large(x, s_out, c_out);
        return;
    }
    const float j = rintf(x * 0.636619747f);
    float r = fmaf(j, -1.57079601e+00f, x);
    r = fmaf(j, -3.13916473e-07f, r);
    r = fmaf(j, -5.39030253e-15f, r);
    sincos_reduced(r, (int)j, s_out, c_out);
}

// ---- a10: read-out activations (layers.py:395-396) ----------------------------------------------
MV_HD float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }
MV_HD float softplus_f32(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }

// ---- a5/a7: row r of a row-major 4x4 times (x,y,z,w), left to right, no FMA --------------------
MV_HD float row_dot4(const float* m, int r, float x, float y, float z, float w) {
    float acc = m[4 * r + 0] * x;
    acc = acc + m[4 * r + 1] * y;
    acc = acc + m[4 * r + 2] * z;
    acc = acc + m[4 * r + 3] * w;
    return acc;
}

struct Taps {
    int x0, y0;      // clamped floor, tl texel; the other taps are (x0+1,y0), (x0,y0+1), (x0+1,y0+1)
    float ax, ay;    // lerp factors in [0,1]
};

// a5 tail + a6 head: q = K4 * cam ; pix = q.xy / max(q.z, 1e-8), clipped to +-1e6 (nerf_utils.py:76-78);
// then tensorflow_addons interpolate_bilinear's floor/alpha with indexing='xy' (border clamp, Q5).
MV_HD void pixel_from_cam(const float* k4, const float cam[4], float* px, float* py) {
    const float q0 = row_dot4(k4, 0, cam[0], cam[1], cam[2], cam[3]);
    const float q1 = row_dot4(k4, 1, cam[0], cam[1], cam[2], cam[3]);
    const float q2 = row_dot4(k4, 2, cam[0], cam[1], cam[2], cam[3]);
    const float den = fmaxf(q2, 1e-8f);
    *px = fminf(fmaxf(q0 / den, -1e6f), 1e6f);
    *py = fminf(fmaxf(q1 / den, -1e6f), 1e6f);
}

MV_HD Taps bilinear_taps(float px, float py, int height, int width) {
    Taps t;
    const float fx = fminf(fmaxf(0.0f, floorf(px)), (float)(width - 2));
    const float fy = fminf(fmaxf(0.0f, floorf(py)), (float)(height - 2));
    t.ax = fminf(fmaxf(0.0f, px - fx), 1.0f);
    t.ay = fminf(fmaxf(0.0f, py - fy), 1.0f);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    return t;
}

// tfa interpolate_bilinear: top = ax*(tr-tl)+tl ; bot = ax*(br-bl)+bl ; out = ay*(bot-top)+top
MV_HD float bilerp(float tl, float tr, float bl, float br, float ax, float ay) {
    const float top = ax * (tr - tl) + tl;
    const float bot = ax * (br - bl) + bl;
    return ay * (bot - top) + top;
}

// a4: stratified edge i in float64 then rounded (nerf_utils.py:50-52), z = lower + u*step32
MV_HD float stratified_z(double near, double far, int n_samples, int i, float u) {
    const double step = (far - near) / (double)n_samples;
    const float lower = (float)(near + (double)i * step);
    const float step32 = (float)step;
    return lower + u * step32;
}

// a11: nerf_utils.py:139
MV_HD float sigma_to_alpha(float sigma, float dist) { return 1.0f - expf(-dist * fmaxf(sigma, 0.0f)); }

}  // namespace mvnerf
