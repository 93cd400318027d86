// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (and the raw TCC counters behind them) on gfx950 against KNOWN byte counts, one
// kernel per access shape the library's kernels use.  Every kernel touches each byte of a buffer far larger than the 256 MiB Infinity
// Cache exactly once, so the expected HBM traffic is the buffer size.  Measurement infrastructure, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/pmc_calibrate tools/pmc_calibrate.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o cal -- tools/bin/pmc_calibrate      (tools/profile_calibrate.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

template <typename T> __device__ inline float fold(T v);
template <> __device__ inline float fold<float>(float v) { return v; }
template <> __device__ inline float fold<float2v>(float2v v) { return v.x + v.y; }
template <> __device__ inline float fold<float4v>(float4v v) { return v.x + v.y + v.z + v.w; }

// coalesced streaming read, sizeof(T) bytes per lane per load, grid-stride
template <typename T>
__global__ __launch_bounds__(256) void cal_read(const T* __restrict__ p, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += fold<T>(p[i]);
    if (acc == 123.456f) sink[0] = acc;
}
// gather of whole 128-B lines in a pseudo-random (bijective) order: 8 consecutive lanes read one line with 16-B loads -- the shape of a
// channel-last bilinear tap (field kernel, ToRGB skip taps)
__global__ __launch_bounds__(256) void cal_gather_line128_b128(const float4v* __restrict__ p, size_t lines, uint64_t mul, float* __restrict__ sink) {
    float acc = 0.f;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t l = t >> 3; l < lines; l += ((size_t)gridDim.x * blockDim.x) >> 3) {
        size_t ln = (l * mul) & (lines - 1);                    // lines is a power of two, mul odd: a bijection
        acc += fold<float4v>(p[ln * 8 + (t & 7)]);
    }
    if (acc == 123.456f) sink[0] = acc;
}
// the same gather with 4 lanes x two 16-B loads per 128-B line is what the walk kernel does per tap (lane = channel quarter)
__global__ __launch_bounds__(256) void cal_gather_line128_2xb128(const float4v* __restrict__ p, size_t lines, uint64_t mul, float* __restrict__ sink) {
    float acc = 0.f;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t l = t >> 2; l < lines; l += ((size_t)gridDim.x * blockDim.x) >> 2) {
        size_t ln = (l * mul) & (lines - 1);
        acc += fold<float4v>(p[ln * 8 + (t & 3) * 2]) + fold<float4v>(p[ln * 8 + (t & 3) * 2 + 1]);
    }
    if (acc == 123.456f) sink[0] = acc;
}
// 64-B half lines gathered at random (4 lanes x 16 B): does a half-line request fetch 64 or 128 B?
__global__ __launch_bounds__(256) void cal_gather_half64_b128(const float4v* __restrict__ p, size_t halves, uint64_t mul, float* __restrict__ sink) {
    float acc = 0.f;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t l = t >> 2; l < halves; l += ((size_t)gridDim.x * blockDim.x) >> 2) {
        size_t h = (l * mul) & (halves - 1);
        acc += fold<float4v>(p[h * 4 + (t & 3)]);
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void cal_write(T* __restrict__ p, size_t n, float v) {
    T x;
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) ((float*)&x)[k] = v;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = x;
}
// read + write of equal size (a copy): the FIR / ToRGB shape
__global__ __launch_bounds__(256) void cal_copy_b128(const float4v* __restrict__ a, float4v* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
    size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;           // MiB; default 2 GiB per buffer
    float *a, *b, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    const uint64_t mul = 0x9E3779B97F4A7C15ull | 1ull;
    for (int rep = 0; rep < 3; ++rep) {
        cal_read<float><<<grid, 256>>>(a, bytes / 4, sink);
        cal_read<float2v><<<grid, 256>>>((const float2v*)a, bytes / 8, sink);
        cal_read<float4v><<<grid, 256>>>((const float4v*)a, bytes / 16, sink);
        cal_gather_line128_b128<<<grid, 256>>>((const float4v*)a, bytes / 128, mul, sink);
        cal_gather_line128_2xb128<<<grid, 256>>>((const float4v*)a, bytes / 128, mul, sink);
        cal_gather_half64_b128<<<grid, 256>>>((const float4v*)a, bytes / 64, mul, sink);
        cal_write<float><<<grid, 256>>>(b, bytes / 4, 1.f);
        cal_write<float4v><<<grid, 256>>>((float4v*)b, bytes / 16, 2.f);
        cal_copy_b128<<<grid, 256>>>((const float4v*)a, (float4v*)b, bytes / 16);
    }
    CK(hipDeviceSynchronize());
    printf("expected_bytes_per_launch %zu\n", bytes);
    return 0;
}
