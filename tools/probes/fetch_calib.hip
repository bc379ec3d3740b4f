// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 by access width (MI355X_MICROARCH.md: FETCH_SIZE reports half the
// bytes of a 16 B/lane streaming read; "other access widths and WRITE_SIZE are uncalibrated").  Each kernel streams a KNOWN number
// of bytes (1 GiB, far past the 256 MiB Infinity Cache) with 4 / 8 / 16 bytes per lane, fully coalesced:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o tools/probes/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- tools/probes/fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- tools/probes/fetch_calib
// tools/pmc_fetch_calibration.py turns the two csv files into bytes-reported / bytes-moved per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename V>
__global__ void read_w(const V* __restrict__ x, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = __builtin_nontemporal_load(x + i);
        acc += ((const float*)&v)[0];
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}
template <typename V>
__global__ void write_w(V* __restrict__ y, size_t n) {
    V v;
    for (int k = 0; k < (int)(sizeof(V) / 4); ++k) ((float*)&v)[k] = 1.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = v;
}
// one wave per 1408-element bf16 row, 8 bytes per lane and pass (the LayerNorm kernels' pattern)
__global__ void read_rows8(const uint2* __restrict__ x, float* __restrict__ out, size_t rows) {
    float acc = 0.f;
    const int lane = threadIdx.x & 63;
    for (size_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4)
        for (int i = lane; i < 352; i += 64) acc += __uint_as_float(x[r * 352 + i].x);
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

int main() {
    const size_t bytes = 1ull << 30;
    void *a, *b;
    float* o;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&o, 4096);
    hipMemset(a, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_w<float>, dim3(8192), dim3(256), 0, 0, (const float*)a, o, bytes / 4);
        hipLaunchKernelGGL(read_w<v2f>, dim3(8192), dim3(256), 0, 0, (const v2f*)a, o, bytes / 8);
        hipLaunchKernelGGL(read_w<v4f>, dim3(8192), dim3(256), 0, 0, (const v4f*)a, o, bytes / 16);
        hipLaunchKernelGGL(read_rows8, dim3(8192), dim3(256), 0, 0, (const uint2*)a, o, bytes / 2816);
        hipLaunchKernelGGL(write_w<float>, dim3(8192), dim3(256), 0, 0, (float*)b, bytes / 4);
        hipLaunchKernelGGL(write_w<v2f>, dim3(8192), dim3(256), 0, 0, (v2f*)b, bytes / 8);
        hipLaunchKernelGGL(write_w<v4f>, dim3(8192), dim3(256), 0, 0, (v4f*)b, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("moved %zu bytes per launch\n", bytes);
    return 0;
}
