// Development only: accuracy (in ulp of the exactly rounded value) of candidate fp32 softplus formulas on the device's own OCML / hardware
// transcendentals.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dev/softplus_acc.hip -o tools/dev/variants/softplus_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ float sp_ocml(float x) { return x > 20.f ? x : log1pf(expf(x)); }
// B: log(1 + y) + (rounding error of 1 + y) / (1 + y)
__device__ float sp_b(float x) {
    const float y = expf(x);
    const float u = 1.0f + y;
    const float err = y <= 1.0f ? y - (u - 1.0f) : 1.0f - (u - y);
    const float r = logf(u) + err * __builtin_amdgcn_rcpf(u);
    return x > 20.f ? x : r;
}
// C: max(x, 0) + log1p(exp(-|x|)), log1p(t) = 2 atanh(t / (2 + t))
__device__ float sp_c(float x) {
    const float t = expf(-fabsf(x));
    const float s = t / (2.0f + t);
    const float z = s * s;
    float p = 1.0f / 17.0f;
    p = fmaf_(p, z, 1.0f / 15.0f);
    p = fmaf_(p, z, 1.0f / 13.0f);
    p = fmaf_(p, z, 1.0f / 11.0f);
    p = fmaf_(p, z, 1.0f / 9.0f);
    p = fmaf_(p, z, 1.0f / 7.0f);
    p = fmaf_(p, z, 1.0f / 5.0f);
    p = fmaf_(p, z, 1.0f / 3.0f);
    const float l1 = fmaf_(2.0f * s * z, p, 2.0f * s);
    const float r = fmaxf(x, 0.0f) + l1;
    return x > 20.f ? x : r;
}
// D: max(x, 0) + [log(1 + t) + err / (1 + t)], t = exp(-|x|) <= 1
__device__ float sp_d(float x) {
    const float t = expf(-fabsf(x));
    const float u = 1.0f + t;
    const float err = t - (u - 1.0f);
    const float l1 = logf(u) + err * __builtin_amdgcn_rcpf(u);
    const float r = fmaxf(x, 0.0f) + l1;
    return x > 20.f ? x : r;
}
__global__ void run(const float* x, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[i] = sp_ocml(x[i]); o[n + i] = sp_b(x[i]); o[2 * n + i] = sp_c(x[i]); o[3 * n + i] = sp_d(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), o(4 * n);
    for (int i = 0; i < n; i++) x[i] = -30.0f + 52.0f * (float)i / (float)n + 1e-3f * (float)(i % 7);
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 4 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    run<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 4 * n * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"ocml log1pf(expf)", "B log(u)+err/u", "C x+ + atanh series", "D x+ + log(u)+err/u"};
    for (int k = 0; k < 4; k++) {
        double mx = 0, sum = 0; float at = 0;
        for (int i = 0; i < n; i++) {
            const long double t = x[i] > 20.f ? (long double)x[i] : log1pl(expl((long double)x[i]));
            const float tr = (float)t;
            const double ulp = (double)(nextafterf(fabsf(tr), INFINITY) - fabsf(tr));
            const double e = fabs((double)o[k * n + i] - (double)t) / ulp;
            sum += e;
            if (e > mx) { mx = e; at = x[i]; }
        }
        printf("%-22s max %.3f ulp (x = %.5f)  mean %.3f ulp\n", names[k], mx, at, sum / n);
    }
    return 0;
}
