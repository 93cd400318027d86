// render_fused.hip -- ImportanceRenderer.forward as ONE C-ABI entry point.
//
// Replaces (one Python method in the reference, ten eager stages): src/training/tri_plane_renderer.py:126-170
//   sample_stratified (:208) -> run_model / simple_tri_plane_renderer + TriPlaneMLP (:172, :560, networks_epigraf.py:46) -> ray marcher on
//   s-space depths (:152) -> sample_importance / sample_pdf (:237, :257) -> run_model on the fine samples -> unify_samples (:196) -> ray
//   marcher in t-space (:166) -> (rgb, depth, weights.sum(2), final_transmittance).
// What a maintainer of the reference binds where `self.renderer(planes, decoder, ray_o, ray_d, rendering_options)` is called
// (networks_epigraf.py:233-240): one call, caller-owned workspace, nothing allocated, nothing synchronised.  The stages are the library's own
// entry points (sampling.hip, field.hip) issued back to back on `stream` -- the same kernels, the same bits as the staged calls
// (tests/test_gpu_parity.py::test_render_fused_equals_staged_calls).
#include "common.h"
#include "../../include/tdgp.h"

namespace {
inline int64_t al256(int64_t v) { return (v + 255) / 256 * 256; }
struct FusedWs { int64_t sdist, tdist, rgbs_c, tfine, rgbs_f, total; };
inline FusedWs fused_ws(int64_t rays, int S, int N) {
    FusedWs w;
    int64_t o = 0;
    w.sdist = o;  o += al256(rays * S * 4);
    w.tdist = o;  o += al256(rays * S * 4);
    w.rgbs_c = o; o += al256(rays * S * 16);
    w.tfine = o;  o += al256(rays * (int64_t)(N > 0 ? N : 1) * 4);
    w.rgbs_f = o; o += al256(rays * (int64_t)(N > 0 ? N : 1) * 16);
    w.total = o;
    return w;
}
}  // namespace

TDGP_API int64_t tdgp_render_fused_workspace_bytes(int B, int64_t R, int S, int N) {
    if (B < 0 || R < 0 || S < 1 || N < 0) return -1;
    return fused_ws((int64_t)B * R, S, N).total;
}

TDGP_API int tdgp_render_fused(const float* planes_hwc, const float* w0, const float* b0, const float* w1, const float* b1,
                               const float* ray_o, const float* ray_d, const float* u_coarse, const float* u_fine,
                               float* rgb, float* depth, float* wsum, float* final_T,
                               int B, int64_t R, int ray_w, int S, int N, int F, int H, int W, int hid, float scale,
                               float t_near, float t_far, int marcher, int flags, float density_bias,
                               void* workspace, int64_t workspace_bytes, tdgp_stream_t stream) {
    TDGP_CHECK(planes_hwc && w0 && b0 && w1 && b1 && ray_o && ray_d && u_coarse && rgb && depth, TDGP_EINVAL, "render_fused: null pointer");
    TDGP_CHECK(N == 0 || u_fine, TDGP_EINVAL, "render_fused: %d fine steps need u_fine", N);
    TDGP_CHECK(B >= 0 && R >= 0 && S >= 2 && N >= 0, TDGP_EINVAL, "render_fused: bad shape (B=%d, R=%lld, S=%d, N=%d)", B, (long long)R, S, N);
    TDGP_CHECK(N > 0, TDGP_EUNSUPPORTED, "render_fused: the generator path always resamples (num_fine_steps = num_ray_steps, networks_epigraf.py:226-231); "
               "N = 0 goes through tdgp_sample_stratified / tdgp_triplane_field / tdgp_ray_march");
    const int64_t rays = (int64_t)B * R;
    if (rays == 0) return TDGP_OK;
    const FusedWs ws = fused_ws(rays, S, N);
    TDGP_CHECK(workspace && workspace_bytes >= ws.total, TDGP_EINVAL, "render_fused: workspace of %lld bytes, need %lld (tdgp_render_fused_workspace_bytes)",
               (long long)workspace_bytes, (long long)ws.total);
    TDGP_CHECK(((uintptr_t)workspace & 15) == 0, TDGP_EINVAL, "render_fused: workspace must be 16-byte aligned");
    char* base = (char*)workspace;
    float* sdist = (float*)(base + ws.sdist);
    float* tdist = (float*)(base + ws.tdist);
    float* rgbs_c = (float*)(base + ws.rgbs_c);
    float* tfine = (float*)(base + ws.tfine);
    float* rgbs_f = (float*)(base + ws.rgbs_f);
    int rc;
    if ((rc = tdgp_sample_stratified(u_coarse, sdist, tdist, rays, S, marcher, t_near, t_far, stream)) != TDGP_OK) return rc;
    if ((rc = tdgp_triplane_field(planes_hwc, nullptr, ray_o, ray_d, tdist, w0, b0, w1, b1, nullptr, 0.f, rgbs_c, nullptr, B, R * S, S, ray_w, F, H, W, hid,
                                  scale, marcher, stream)) != TDGP_OK) return rc;
    if ((rc = tdgp_importance_from_coarse(rgbs_c, sdist, u_fine, tfine, nullptr, nullptr, nullptr, rays, S, N, marcher, flags, density_bias, 0.f, t_near, t_far,
                                          stream)) != TDGP_OK) return rc;
    if ((rc = tdgp_triplane_field(planes_hwc, nullptr, ray_o, ray_d, tfine, w0, b0, w1, b1, nullptr, 0.f, rgbs_f, nullptr, B, R * N, N, ray_w, F, H, W, hid,
                                  scale, marcher, stream)) != TDGP_OK) return rc;
    return tdgp_merge_composite(rgbs_c, tdist, S, rgbs_f, tfine, N, rgb, depth, wsum, final_T, nullptr, nullptr, rays, marcher, flags, density_bias, 0.f, stream);
}
