// extern "C" shims over mvnerf_math.h so tests/test_device_math_cpu.py can call the scalar code.
#include "../../thesis_clip_nerf_amd/csrc/mvnerf_math.h"

extern "C" {
void mv_sincos(const float* x, int n, float* s, float* c) {
    for (int i = 0; i < n; ++i) mvnerf::sincos_f32(x[i], &s[i], &c[i]);
}
void mv_sigmoid_softplus(const float* x, int n, float* sg, float* sp) {
    for (int i = 0; i < n; ++i) {
        sg[i] = mvnerf::sigmoid_f32(x[i]);
        sp[i] = mvnerf::softplus_f32(x[i]);
    }
}
// cam (n,4), k4 (16) -> pix (n,2), taps x0,y0 (n) and ax, ay (n)
void mv_project(const float* k4, const float* cam, int n, int height, int width, float* pix, int* x0, int* y0,
                float* ax, float* ay) {
    for (int i = 0; i < n; ++i) {
        float px, py;
        mvnerf::pixel_from_cam(k4, cam + 4 * i, &px, &py);
        const mvnerf::Taps t = mvnerf::bilinear_taps(px, py, height, width);
        pix[2 * i] = px; pix[2 * i + 1] = py;
        x0[i] = t.x0; y0[i] = t.y0; ax[i] = t.ax; ay[i] = t.ay;
    }
}
void mv_matvec_rows(const float* m, const float* xyz, int n, float w, float* out) {
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 4; ++r) out[4 * i + r] = mvnerf::row_dot4(m, r, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], w);
}
void mv_stratified(const float* u, int n_rays, int n_samples, double near_, double far_, float* z) {
    for (int i = 0; i < n_rays * n_samples; ++i) z[i] = mvnerf::stratified_z(near_, far_, n_samples, i % n_samples, u[i]);
}
void mv_bilerp(const float* t, int n, float* out) {   // t: (n,6) = tl,tr,bl,br,ax,ay
    for (int i = 0; i < n; ++i) out[i] = mvnerf::bilerp(t[6 * i], t[6 * i + 1], t[6 * i + 2], t[6 * i + 3], t[6 * i + 4], t[6 * i + 5]);
}
}
