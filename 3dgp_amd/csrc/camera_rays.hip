// camera_rays.hip -- cam2world matrices and per-pixel rays.
//
// Replaces (Python/eager in the reference):
//   src/training/rendering_utils.py:194-218  compute_cam2world_matrix (+ :270-285 spherical2cartesian,
//                                             :28-32 normalize without epsilon)
//   src/training/tri_plane_renderer.py:487-527  sample_rays   (closed form: SURVEY.md 10.4)
// Both are tiny (B matrices, 24 B per ray); they exist as kernels so the whole forward stays on the
// stream without host round trips.  Transcendentals are evaluated in fp64 and rounded once.
#include "common.h"

namespace {

__device__ __forceinline__ void sph2cart(float rot, float pitch, float radius, float* o) {
    float sp = (float)sin((double)pitch), cp = (float)cos((double)pitch);
    float sr = (float)sin((double)(-rot)), cr = (float)cos((double)rot);
    o[0] = radius * sp * sr;
    o[1] = radius * cp;
    o[2] = radius * sp * cr;
}
// The fp32 chains below follow what the reference's CPU/PyTorch path executes (probed against torch 2.10; DESIGN.md (c), "fp32 chains restated"):
// torch.norm over 3 components = x0*x0 -> fma -> fma -> sqrt; torch.cross = fma(a_i, b_j, -(a_j * b_i)); the K = 3 bmm = an in-order
// fused chain.  With these the rays are bit-identical to the reference's whenever the cam2world matrix is (its sin / cos are Sleef's
// 1-ulp routines there, correctly rounded here: ~5 % of angles differ in the last bit).
__device__ __forceinline__ void norm3(float* v) {
    // sqrtf and `/` are the correctly rounded fp32 forms in this build (hipcc's default; `__fsqrt_rn` is NOT -- it maps to the native sqrt)
    float n = sqrtf(__fmaf_rn(v[2], v[2], __fmaf_rn(v[1], v[1], v[0] * v[0])));
    v[0] = v[0] / n; v[1] = v[1] / n; v[2] = v[2] / n;
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = __fmaf_rn(a[1], b[2], -(a[2] * b[1]));
    o[1] = __fmaf_rn(a[2], b[0], -(a[0] * b[2]));
    o[2] = __fmaf_rn(a[0], b[1], -(a[1] * b[0]));
}

__global__ void cam2world_kernel(const float* angles, const float* radius, const float* look_at, float* c2w, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float org[3], la[3], fwd[3], left[3], upv[3];
    const float up0[3] = {0.f, 1.f, 0.f};
    sph2cart(angles[b * 3 + 0], angles[b * 3 + 1], radius[b], org);
    sph2cart(look_at[b * 3 + 0], look_at[b * 3 + 1], look_at[b * 3 + 2], la);
    for (int i = 0; i < 3; i++) fwd[i] = la[i] - org[i];
    norm3(fwd); norm3(fwd);                       // normalised twice: rendering_utils.py:204,206
    cross3(up0, fwd, left); norm3(left);
    cross3(fwd, left, upv); norm3(upv);
    float* m = c2w + b * 16;
    for (int r = 0; r < 3; r++) {
        m[r * 4 + 0] = -left[r];
        m[r * 4 + 1] = upv[r];
        m[r * 4 + 2] = -fwd[r];
        m[r * 4 + 3] = org[r];
    }
    m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}

// torch.linspace (fp32, the CPU kernel the reference path runs): symmetric fill around the midpoint, each element ONE fused
// multiply-add (ATen's serial fill `start + step * idx` / `end - step * (steps - idx - 1)` is compiled with contraction; pinned by
// the full-size reference goldens, tests/golden/e2e_full_*.npz -- the two-rounding form is off by an ulp on 124 of 256 columns).
__device__ __forceinline__ float linspace_f(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? __fmaf_rn(step, (float)i, start) : __fmaf_rn(-step, (float)(steps - 1 - i), end);
}

__global__ __launch_bounds__(256) void sample_rays_kernel(const float* __restrict__ c2w, const float* __restrict__ fov, int fov_stride,
                                                         const float* __restrict__ ps, const float* __restrict__ po,
                                                         float* __restrict__ ray_o, float* __restrict__ ray_d, int B, int h, int w) {
    const int64_t total = (int64_t)B * h * w;
    const float pi_f = (float)3.141592653589793;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < total; r += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(r % w);
        int i = (int)((r / w) % h);
        int b = (int)(r / ((int64_t)w * h));
        const float* m = c2w + b * 16;
        float fov_rad = fov[b * fov_stride] / 360.f * 2.f * pi_f;
        float z = -1.0f / (float)tan((double)(fov_rad * 0.5f));
        float x = linspace_f(-1.f, 1.f, w, j);
        float y = linspace_f(1.f, -1.f, h, i);
        if (ps) {
            x = (x + 1.0f) * ps[b * 2 + 0] - 1.0f + po[b * 2 + 0] * 2.0f;
            y = (y + 1.0f) * ps[b * 2 + 1] - 1.0f + po[b * 2 + 1] * 2.0f;
        }
        float d[3] = {x, y, z};
        norm3(d);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ray_d[r * 3 + k] = __fmaf_rn(m[k * 4 + 2], d[2], __fmaf_rn(m[k * 4 + 1], d[1], m[k * 4 + 0] * d[0]));
            ray_o[r * 3 + k] = m[k * 4 + 3];
        }
    }
}

__global__ __launch_bounds__(256) void rays_to_image_kernel(const float* __restrict__ rgb, float* __restrict__ img, int B, int hw) {
    const int64_t total = (int64_t)B * 3 * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int p = (int)(i % hw);
        int c = (int)((i / hw) % 3);
        int b = (int)(i / ((int64_t)3 * hw));
        img[i] = rgb[((int64_t)b * hw + p) * 3 + c];
    }
}

}  // namespace

TDGP_API int tdgp_cam2world(const float* angles, const float* radius, const float* look_at, float* c2w, int B, tdgp_stream_t stream) {
    TDGP_CHECK(angles && radius && look_at && c2w, TDGP_EINVAL, "cam2world: null pointer");
    TDGP_CHECK(B >= 0, TDGP_EINVAL, "cam2world: negative batch");
    if (B == 0) return TDGP_OK;
    TDGP_LAUNCH("cam2world_kernel", cam2world_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, angles, radius, look_at, c2w, B);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_sample_rays(const float* c2w, const float* fov, int fov_stride, const float* patch_scales,
                              const float* patch_offsets, float* ray_o, float* ray_d, int B, int h, int w, tdgp_stream_t stream) {
    TDGP_CHECK(c2w && fov && ray_o && ray_d, TDGP_EINVAL, "sample_rays: null pointer");
    TDGP_CHECK((patch_scales == nullptr) == (patch_offsets == nullptr), TDGP_EINVAL, "sample_rays: patch scales/offsets must come together");
    TDGP_CHECK(B >= 0 && h >= 1 && w >= 1, TDGP_EINVAL, "sample_rays: bad shape");
    TDGP_CHECK(fov_stride == 0 || fov_stride == 1, TDGP_EINVAL, "sample_rays: fov_stride must be 0 or 1");
    if (B == 0) return TDGP_OK;
    const int64_t total = (int64_t)B * h * w;
    TDGP_LAUNCH("sample_rays_kernel", sample_rays_kernel, dim3((int)min((int64_t)4096, cdiv64(total, 256))), dim3(256), 0, (hipStream_t)stream, c2w, fov,
                       fov_stride, patch_scales, patch_offsets, ray_o, ray_d, B, h, w);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_rays_to_image(const float* rgb, float* img, int B, int hw, tdgp_stream_t stream) {
    TDGP_CHECK(rgb && img, TDGP_EINVAL, "rays_to_image: null pointer");
    if (B == 0 || hw == 0) return TDGP_OK;
    const int64_t total = (int64_t)B * 3 * hw;
    TDGP_LAUNCH("rays_to_image_kernel", rays_to_image_kernel, dim3((int)min((int64_t)4096, cdiv64(total, 256))), dim3(256), 0, (hipStream_t)stream, rgb, img, B, hw);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
