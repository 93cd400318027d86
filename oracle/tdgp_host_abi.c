/*
 * tdgp_host_abi.c -- a HOST (CPU memory) build of the two C-ABI entry points that replace the reference's pybind plugins:
 *     tdgp_bias_act   <->  src/torch_utils/ops/bias_act.cpp:32
 *     tdgp_upfirdn2d  <->  src/torch_utils/ops/upfirdn2d.cpp:16
 * with exactly the prototypes of include/tdgp.h (pointers are host pointers, the stream argument is ignored).
 *
 * TEST INFRASTRUCTURE ONLY (it lives under oracle/ for that reason).  Purpose: in the build container, where there is
 * no GPU, tests/test_compat.py runs the REFERENCE's own model code -- networks_stylegan2.py, layers.py, tri_plane_renderer.py -- with
 * this package's op modules shadowing src.torch_utils.ops and THIS library loaded in the place of libtdgp_hip.so (TDGP_LIB_PATH), so
 * that every bias_act / upfirdn2d the reference issues travels  custom_ops plugin object -> ctypes -> tdgp_* entry point  and the
 * generator's golden image must come out.  That proves the boundary (names, argument order and meaning, strides, error codes) from
 * the reference's call sites, not just the Python signatures.  The product never loads this file: 3dgp_amd/ has no CPU path.
 *
 * Every other entry point of the ABI is exported as a stub that fails with TDGP_EUNSUPPORTED (generated: _build/tdgp_host_stubs.c), so the
 * loader's "all symbols present" check holds.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../include/tdgp.h"

#define HOST_API __attribute__((visibility("default")))

static __thread char g_err[256] = "";
HOST_API const char* tdgp_last_error(void) { return g_err; }
HOST_API int tdgp_version(void) { return 100; }

static float host_act(float x, int act, float alpha)
{
    switch (act) {                     /* bias_act.py:21-31, ids = cuda_idx */
    case 1: return x;
    case 2: return x > 0.f ? x : 0.f;
    case 3: return x > 0.f ? x : x * alpha;
    case 4: return tanhf(x);
    case 5: return 1.f / (1.f + expf(-x));
    case 6: return x > 0.f ? x : expm1f(x);
    case 7: return x > 0.f ? 1.0507009873554804934193349852946f * x : 1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f * expm1f(x);
    case 8: return x > 20.f ? x : log1pf(expf(x));
    case 9: return x / (1.f + expf(-x));
    }
    return x;
}

HOST_API int tdgp_bias_act(const void* x, const void* b, void* y, int64_t n, int sizeB, int64_t stepB, int act, float alpha, float gain,
                           float clamp, int dtype, tdgp_stream_t stream)
{
    (void)stream;
    if (!x || !y) { snprintf(g_err, sizeof g_err, "bias_act: null pointer"); return TDGP_EINVAL; }
    if (dtype != TDGP_F32) { snprintf(g_err, sizeof g_err, "bias_act (host build): fp32 only"); return TDGP_EUNSUPPORTED; }
    if (act < 1 || act > 9) { snprintf(g_err, sizeof g_err, "bias_act: unknown activation %d", act); return TDGP_EUNSUPPORTED; }
    const float* xp = (const float*)x; const float* bp = (const float*)b; float* yp = (float*)y;
    for (int64_t i = 0; i < n; i++) {
        float v = xp[i];
        if (bp) v = v + bp[(i / stepB) % sizeB];
        v = host_act(v, act, alpha) * gain;
        if (clamp >= 0.f) v = v < -clamp ? -clamp : (v > clamp ? clamp : v);
        yp[i] = v;
    }
    return TDGP_OK;
}

HOST_API int tdgp_upfirdn2d(const void* x, const float* f, void* y, int N, int C, int inH, int inW, const int64_t* xs, int outH, int outW,
                            const int64_t* ys, int fH, int fW, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                            int flip, float gain, int dtype, tdgp_stream_t stream)
{
    (void)stream; (void)padx1; (void)pady1;
    if (!x || !f || !y || !xs || !ys) { snprintf(g_err, sizeof g_err, "upfirdn2d: null pointer"); return TDGP_EINVAL; }
    if (dtype != TDGP_F32) { snprintf(g_err, sizeof g_err, "upfirdn2d (host build): fp32 only"); return TDGP_EUNSUPPORTED; }
    if (upx < 1 || upy < 1 || downx < 1 || downy < 1 || outH < 1 || outW < 1) { snprintf(g_err, sizeof g_err, "upfirdn2d: bad geometry"); return TDGP_EINVAL; }
    const float* xp = (const float*)x; float* yp = (float*)y;
    /* out[oy,ox] = gain * sum_{fy,fx} F[fy,fx] * U[oy*downy + fy - pady0, ox*downx + fx - padx0],  U = zero-stuffed x,
     * F = the filter flipped unless `flip` (upfirdn2d.cpp / upfirdn2d.py:167-211) */
    for (int n = 0; n < N; n++)
        for (int c = 0; c < C; c++)
            for (int oy = 0; oy < outH; oy++)
                for (int ox = 0; ox < outW; ox++) {
                    double acc = 0.0;
                    for (int fy = 0; fy < fH; fy++) {
                        const int uy = oy * downy + fy - pady0;
                        if (uy < 0 || uy % upy) continue;
                        const int iy = uy / upy;
                        if (iy >= inH) continue;
                        for (int fx = 0; fx < fW; fx++) {
                            const int ux = ox * downx + fx - padx0;
                            if (ux < 0 || ux % upx) continue;
                            const int ix = ux / upx;
                            if (ix >= inW) continue;
                            const float w = flip ? f[fy * fW + fx] : f[(fH - 1 - fy) * fW + (fW - 1 - fx)];
                            acc += (double)w * (double)xp[n * xs[0] + c * xs[1] + iy * xs[2] + ix * xs[3]];
                        }
                    }
                    yp[n * ys[0] + c * ys[1] + oy * ys[2] + ox * ys[3]] = (float)acc * gain;
                }
    return TDGP_OK;
}

/* the generated stubs (a separate translation unit: they do not see the header's prototypes) report through this */
void tdgp_host_set_error(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
