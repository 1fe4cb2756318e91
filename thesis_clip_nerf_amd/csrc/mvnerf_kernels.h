// Internal launch interface between api.hip and the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvnerf {

struct FieldParams {
    const float* rays_o;
    const float* rays_d;
    const float* z;
    const float* images;
    const float* features;
    const float* k4;
    const float* einv;
    const float* net;      // packed (mvnerf_pack.h)
    float* rgbs;
    int32_t* tap_idx;      // optional
    float* pix;            // optional
    int B, V, R, S, H, W;
    long total;            // B*R*S samples
    long n_tiles;          // ceil(total / 32)
    unsigned int* tile_counter;   // set by launch_field_eval
};

hipError_t launch_pack_net(const float* net_keras, float* packed, hipStream_t stream);
hipError_t launch_field_eval(const FieldParams& p, hipStream_t stream);

hipError_t launch_get_rays(const double* m9, const double* origin3, const float* u, const float* v, int n_rays,
                           int width, int normalize, float* rays_o, float* rays_d, double* rays_d64,
                           hipStream_t stream);
hipError_t launch_stratified(const float* u, long n, int n_samples, double near_, double far_, float* z,
                             hipStream_t stream);
hipError_t launch_composite(const float* z, const float* rgbs, int n_rays, int S, float* rgb, float* depth,
                            float* weights, hipStream_t stream);
hipError_t launch_resample(const float* z, const float* weights, const float* u_fine, int n_rays, int q7_mode,
                           float* z_all, float* z_fine, int32_t* above, int32_t* below, hipStream_t stream);

}  // namespace mvnerf
