// What do streaming kernels shaped like the tracker's key LayerNorms reach on gfx950 when there is NO arithmetic in them?
// (round 5: the key LayerNorms sit at 2.4 - 3.0 TB/s of unique traffic, LayerNorm3d at 3.3; a float4 copy is quoted at 6.3)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/stream_bw.hip -o tools/probes/stream_bw && tools/probes/stream_bw
// Each case moves `rows` rows of 1408 elements: R streams of 2-byte elements read, W streams of 2-byte elements written
// (+ optionally 2 shared float rows of period 2048 read per row, the positional / shared-key operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave per row; lane owns 4 consecutive elements per slot (8-byte accesses on 2-byte streams: the LayerNorm kernels' layout)
template <int R, int W, int SH, int ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void rows8(const u32x2* __restrict__ in, u32x2* __restrict__ out, const f32x4* __restrict__ sh, long long rows,
                                            long long stride /* u32x2 per stream */) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    u32x2 v[ROWS_PER_WAVE][R > 0 ? R : 1][6];
    f32x4 s[ROWS_PER_WAVE][SH > 0 ? SH : 1][6];
#pragma unroll
    for (int q = 0; q < ROWS_PER_WAVE; ++q) {
        const long long row = row0 + q;
        if (row >= rows) break;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int idx = lane + 64 * i;
                v[q][r][i] = in[r * stride + row * 352 + (idx < 352 ? idx : 0)];
            }
#pragma unroll
        for (int r = 0; r < SH; ++r)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int idx = lane + 64 * i;
                s[q][r][i] = sh[((long long)r * 2048 + row % 2048) * 352 + (idx < 352 ? idx : 0)];
            }
    }
#pragma unroll
    for (int q = 0; q < ROWS_PER_WAVE; ++q) {
        const long long row = row0 + q;
        if (row >= rows) break;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = lane + 64 * i;
            u32x2 a = {(unsigned)idx, (unsigned)row};
#pragma unroll
            for (int r = 0; r < R; ++r) a.x ^= v[q][r][i].x, a.y += v[q][r][i].y;
#pragma unroll
            for (int r = 0; r < SH; ++r) a.x ^= __float_as_uint(s[q][r][i][0] + s[q][r][i][3]), a.y += __float_as_uint(s[q][r][i][1] * s[q][r][i][2]);
            if (idx < 352) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    u32x2 o = a;
                    o.x += w;
                    out[w * stride + row * 352 + idx] = o;
                }
                if (W == 0 && a.x == 0x12345u && a.y == 0x6789u) out[0] = a;  // (keeps the loads of a read-only case alive)
            }
        }
    }
}
// wave per row; lane owns 8 consecutive elements per slot (16-byte accesses): 176 slots per row = 2.75 passes
template <int R, int W>
__global__ __launch_bounds__(256) void rows16(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long rows, long long stride) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    u32x4 v[R > 0 ? R : 1][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = lane + 64 * i;
            v[r][i] = in[r * stride + row * 176 + (idx < 176 ? idx : 0)];
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = lane + 64 * i;
        u32x4 a = {(unsigned)idx, (unsigned)row, 1u, 2u};
#pragma unroll
        for (int r = 0; r < R; ++r) a.x ^= v[r][i].x, a.y += v[r][i].y, a.z ^= v[r][i].z, a.w += v[r][i].w;
        if (idx < 176) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                u32x4 o = a;
                o.x += w;
                out[w * stride + row * 176 + idx] = o;
            }
            if (W == 0 && a.x == 0x12345u && a.y == 0x6789u) out[0] = a;
        }
    }
}
// flat grid-stride copy, 16 bytes per lane: R input streams combined into W output streams
template <int R, int W>
__global__ __launch_bounds__(256) void flat16(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long n, long long stride) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        u32x4 a = {1u, 2u, 3u, 4u};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u32x4 v = in[r * stride + i];
            a.x ^= v.x, a.y += v.y, a.z ^= v.z, a.w += v.w;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            u32x4 o = a;
            o.x += w;
            out[w * stride + i] = o;
        }
        if (W == 0 && a.x == 0x12345u && a.y == 0x6789u) out[0] = a;
    }
}

static hipEvent_t e0, e1;
template <class F>
static void timeit(const char* name, double unique_bytes, F launch) {
    launch();
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-78s %8.1f us (best %8.1f)  %6.2f TB/s unique\n", name, sum / reps * 1e3, best * 1e3, unique_bytes / (sum / reps * 1e-3) / 1e12);
}

int main() {
    const long long rows = 131072, C = 1408;
    const long long sbytes = rows * C * 2;  // one 2-byte stream: 369 MB
    char *in, *out;
    float* sh;
    hipMalloc(&in, 3 * sbytes);
    hipMalloc(&out, 2 * sbytes);
    hipMalloc(&sh, 2ll * 2048 * C * 4);
    hipMemset(in, 1, 3 * sbytes);
    hipMemset(sh, 0, 2ll * 2048 * C * 4);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long long s8 = sbytes / 8, s16 = sbytes / 16;
    const dim3 grid((rows + 3) / 4), blk(256);
    printf("rows %lld x %lld 2-byte elements: %.0f MB per stream\n", rows, C, sbytes / 1e6);
#define ROWS8(R, W, SH, RPW, label) \
    timeit(label, (double)(R + W) * sbytes, [&] { hipLaunchKernelGGL((rows8<R, W, SH, RPW>), dim3((rows / RPW + 3) / 4), blk, 0, 0, (const u32x2*)in, (u32x2*)out, (const f32x4*)sh, rows, s8); })
    ROWS8(1, 1, 0, 1, "wave/row 8 B/lane: 1 read, 1 write");
    ROWS8(1, 2, 0, 1, "wave/row 8 B/lane: 1 read, 2 writes            (layer-0 key LayerNorm, unique part)");
    ROWS8(2, 2, 0, 1, "wave/row 8 B/lane: 2 reads, 2 writes           (chained key LayerNorm, unique part)");
    ROWS8(1, 2, 2, 1, "wave/row 8 B/lane: 1 read, 2 writes + 2 shared float rows");
    ROWS8(2, 2, 2, 1, "wave/row 8 B/lane: 2 reads, 2 writes + 2 shared float rows");
    ROWS8(2, 2, 0, 2, "2 rows/wave 8 B/lane: 2 reads, 2 writes");
    ROWS8(2, 0, 0, 1, "wave/row 8 B/lane: 2 reads only");
    ROWS8(0, 2, 0, 1, "wave/row 8 B/lane: 2 writes only");
#define ROWS16(R, W, label) \
    timeit(label, (double)(R + W) * sbytes, [&] { hipLaunchKernelGGL((rows16<R, W>), grid, blk, 0, 0, (const u32x4*)in, (u32x4*)out, rows, s16); })
    ROWS16(1, 1, "wave/row 16 B/lane: 1 read, 1 write");
    ROWS16(1, 2, "wave/row 16 B/lane: 1 read, 2 writes");
    ROWS16(2, 2, "wave/row 16 B/lane: 2 reads, 2 writes");
    ROWS16(0, 2, "wave/row 16 B/lane: 2 writes only");
#define FLAT(R, W, G, label) \
    timeit(label, (double)(R + W) * sbytes, [&] { hipLaunchKernelGGL((flat16<R, W>), dim3(G), blk, 0, 0, (const u32x4*)in, (u32x4*)out, s16, s16); })
    FLAT(1, 1, 8192, "flat grid-stride 16 B/lane (8192 blocks): 1 read, 1 write");
    FLAT(2, 2, 8192, "flat grid-stride 16 B/lane (8192 blocks): 2 reads, 2 writes");
    FLAT(1, 2, 8192, "flat grid-stride 16 B/lane (8192 blocks): 1 read, 2 writes");
    FLAT(2, 2, 2048, "flat grid-stride 16 B/lane (2048 blocks): 2 reads, 2 writes");
    FLAT(2, 0, 8192, "flat grid-stride 16 B/lane (8192 blocks): 2 reads only");
    FLAT(0, 2, 8192, "flat grid-stride 16 B/lane (8192 blocks): 2 writes only");
    // the in-place pattern of LayerNorm3d: 1M rows x 352 elements, read and written in place
    timeit("in place, flat 16 B/lane: 738 MB read and written back", 2.0 * 2 * sbytes,
           [&] { hipLaunchKernelGGL((flat16<1, 1>), dim3(8192), blk, 0, 0, (const u32x4*)in, (u32x4*)in, 2 * s16, 0ll); });
    return 0;
}
