// Instrumented victim for the stream-race diagnosis (tools/probes/race_probe.py).
// The seam alignment's pointmap kernel (csrc/umeyama.hip) was seen to compute with ZEROS in one pose row for the last quarter of
// a wave while the tracker recursion ran on another stream.  This kernel repeats exactly that access - every lane of a wave
// loads the same 16-byte row of a small, never-written [F][16] table - and checks the value IN the kernel, three ways:
//   a: global_load_dwordx4            (what the compiler emitted for the victim)
//   b: the same load again            (L1 hit if a's line is still there: was the LINE bad or only a's return?)
//   c: global_load_dwordx4 sc0 sc1    (bypasses the vector L1 / L2 hit path)
// A mismatch is logged with the hardware id of the wave (XCC, SE, CU, SIMD), iteration and lane.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/race_victim.hip -o tools/probes/librace_victim.so
#include <hip/hip_runtime.h>

typedef float f4 __attribute__((ext_vector_type(4)));

struct Rec {
    unsigned launch, iter, gid, row, hwid, xcc, flags, pad;
    float a[4], b[4], c[4];
    unsigned long long t;
    unsigned long long addr;
};

__global__ void victim_kernel(const float* __restrict__ P, int F, int spf, int iters, unsigned launch, Rec* __restrict__ log,
                              unsigned* __restrict__ nlog, unsigned cap, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * spf) return;
    const int f = i / spf;
    const float* p = P + f * 16;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* q = p + 4 * r;
            f4 a, b, c;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(b) : "v"(q) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(q) : "memory");
            // the table holds f + identity rows: row r = e_r scaled by (f + 1)
            const float d = (float)(f + 1);
            const f4 e = {r == 0 ? d : 0.f, r == 1 ? d : 0.f, r == 2 ? d : 0.f, 0.f};
            const bool ba = a.x != e.x || a.y != e.y || a.z != e.z || a.w != e.w;
            const bool bb = b.x != e.x || b.y != e.y || b.z != e.z || b.w != e.w;
            const bool bc = c.x != e.x || c.y != e.y || c.z != e.z || c.w != e.w;
            if (ba || bb || bc) {
                const unsigned slot = atomicAdd(nlog, 1u);
                if (slot < cap) {
                    Rec& R = log[slot];
                    R.launch = launch, R.iter = it, R.gid = i, R.row = r;
                    R.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
                    R.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
                    R.flags = (ba ? 1u : 0u) | (bb ? 2u : 0u) | (bc ? 4u : 0u);
                    R.a[0] = a.x, R.a[1] = a.y, R.a[2] = a.z, R.a[3] = a.w;
                    R.b[0] = b.x, R.b[1] = b.y, R.b[2] = b.z, R.b[3] = b.w;
                    R.c[0] = c.x, R.c[1] = c.y, R.c[2] = c.z, R.c[3] = c.w;
                    R.t = __builtin_readcyclecounter();
                    R.addr = (unsigned long long)q;
                }
            }
            acc += a.x + b.y + c.z;
        }
    }
    out[i] = acc;
}

extern "C" int race_victim_launch(void* stream, const float* P, int F, int spf, int iters, unsigned launch, void* log, unsigned* nlog,
                                  unsigned cap, float* out) {
    hipLaunchKernelGGL(victim_kernel, dim3((F * spf + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, F, spf, iters, launch,
                       (Rec*)log, nlog, cap, out);
    return (int)hipGetLastError();
}

// private device memory outside torch's caching allocator (does the table's allocator matter?)
extern "C" void* race_hip_malloc(size_t n) {
    void* p = nullptr;
    return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
}
extern "C" int race_hip_memcpy_h2d(void* dst, const void* src, size_t n) { return (int)hipMemcpy(dst, src, n, hipMemcpyHostToDevice); }

// The compiler's own instruction sequence for the three pose-row loads of pointmap_kernel (csrc/umeyama.hip at round 4), register
// for register: rows 1, 0, 2 requested back to back from ONE address pair, the last load's destination overlapping that pair.
__global__ void victim_seq_kernel(const float* __restrict__ P, int F, int spf, int iters, unsigned launch, Rec* __restrict__ log,
                                  unsigned* __restrict__ nlog, unsigned cap, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * spf) return;
    const int f = i / spf;
    const float* p = P + f * 16;
    const unsigned lo = (unsigned)(unsigned long long)p, hi = (unsigned)((unsigned long long)p >> 32);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float r[12];
        asm volatile(
            "v_mov_b32 v24, %12\n\tv_mov_b32 v25, %13\n\t"
            "global_load_dwordx4 v[16:19], v[24:25], off offset:16\n\t"
            "global_load_dwordx4 v[20:23], v[24:25], off\n\t"
            "global_load_dwordx4 v[24:27], v[24:25], off offset:32\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23\n\t"
            "v_mov_b32 %4, v16\n\tv_mov_b32 %5, v17\n\tv_mov_b32 %6, v18\n\tv_mov_b32 %7, v19\n\t"
            "v_mov_b32 %8, v24\n\tv_mov_b32 %9, v25\n\tv_mov_b32 %10, v26\n\tv_mov_b32 %11, v27"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
              "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11])
            : "v"(lo), "v"(hi)
            : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
        const float d = (float)(f + 1);
        unsigned badrows = 0;
#pragma unroll
        for (int row = 0; row < 3; ++row)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (r[row * 4 + c] != ((c == row) ? d : 0.f)) badrows |= 1u << row;
        if (badrows) {
            const unsigned slot = atomicAdd(nlog, 1u);
            if (slot < cap) {
                Rec& R = log[slot];
                R.launch = launch, R.iter = it, R.gid = i, R.row = badrows;
                R.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                R.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                R.flags = 8u;
                for (int c = 0; c < 4; ++c) R.a[c] = r[c], R.b[c] = r[4 + c], R.c[c] = r[8 + c];
                R.t = __builtin_readcyclecounter();
                R.addr = (unsigned long long)p;
            }
        }
        acc += r[0] + r[5] + r[10];
    }
    out[i] = acc;
}

extern "C" int race_victim_seq_launch(void* stream, const float* P, int F, int spf, int iters, unsigned launch, void* log,
                                      unsigned* nlog, unsigned cap, float* out) {
    hipLaunchKernelGGL(victim_seq_kernel, dim3((F * spf + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, F, spf, iters, launch,
                       (Rec*)log, nlog, cap, out);
    return (int)hipGetLastError();
}

// The pointmap kernel itself (same source as csrc/umeyama.hip, so the compiler schedules its seven loads with counted vmcnt waits
// as it does there), followed by a second read of every input after a full drain (sc0 sc1, vmcnt(0)): which VALUE THE KERNEL
// COMPUTED WITH differs from what memory holds?  flags: bits 0-2 pose rows, bits 4-6 K rows, bit 7 depth.
__device__ __forceinline__ unsigned hash_u32(unsigned x) {
    x = x * 747796405u + 2891336453u;
    const unsigned w = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ f4 reload4(const float* q) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
    return v;
}
template <int DRAIN>
__global__ void victim_full_kernel(const float* __restrict__ depth, const float* __restrict__ K, const float* __restrict__ P,
                                   float* __restrict__ out, int F, int H, int W, int ratio, unsigned seed, int spf, unsigned launch,
                                   Rec* __restrict__ log, unsigned* __restrict__ nlog, unsigned cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * spf) return;
    const int f = i / spf, j = i % spf;
    int pix = j * ratio + (int)(hash_u32(seed ^ (unsigned)j * 2654435761u) % (unsigned)ratio);
    if (pix >= H * W) pix = H * W - 1;
    const float x = (float)(pix % W), y = (float)(pix / W);
    const float* k = K + f * 16;
    const float a = k[0], b = k[1], c = k[2], d = k[4], e = k[5], g = k[6], h = k[8], l = k[9], m = k[10];
    const float det = a * (e * m - g * l) - b * (d * m - g * h) + c * (d * l - e * h);
    const float id = 1.f / det;
    const float i00 = (e * m - g * l) * id, i01 = (c * l - b * m) * id, i02 = (b * g - c * e) * id;
    const float i10 = (g * h - d * m) * id, i11 = (a * m - c * h) * id, i12 = (c * d - a * g) * id;
    const float i20 = (d * l - e * h) * id, i21 = (b * h - a * l) * id, i22 = (a * e - b * d) * id;
    const float z = depth[(long long)f * H * W + pix];
    const float cx = (i00 * x + i01 * y + i02) * z, cy = (i10 * x + i11 * y + i12) * z, cz = (i20 * x + i21 * y + i22) * z;
    const float* p = P + f * 16;
    const float p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6], p7 = p[7], p8 = p[8], p9 = p[9],
                p10 = p[10], p11 = p[11];
    if (DRAIN == 1) {
        // every load drained before the first use of a pose value (the compiler's own counted waits follow and are no-ops)
        float q0 = p0, q1 = p1, q4 = p4, q5 = p5, q8 = p8;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q4), "+v"(q5), "+v"(q8) : : "memory");
        const float ox = q0 * cx + q1 * cy + p2 * cz + p3, oy = q4 * cx + q5 * cy + p6 * cz + p7, oz = q8 * cx + p9 * cy + p10 * cz + p11;
        out[i * 3 + 0] = ox;
        out[i * 3 + 1] = oy;
        out[i * 3 + 2] = oz;
        return;
    }
    if (DRAIN == 2) {
        // no drain, but a few idle cycles between the compiler's counted wait and the first use (the asm sits behind the loads in
        // program order, so the counted wait for these registers is placed in front of it)
        float q0 = p0, q1 = p1, q4 = p4, q5 = p5, q8 = p8;
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(q0), "+v"(q1), "+v"(q4), "+v"(q5), "+v"(q8) : : "memory");
        const float ox = q0 * cx + q1 * cy + p2 * cz + p3, oy = q4 * cx + q5 * cy + p6 * cz + p7, oz = q8 * cx + p9 * cy + p10 * cz + p11;
        out[i * 3 + 0] = ox;
        out[i * 3 + 1] = oy;
        out[i * 3 + 2] = oz;
        return;
    }
    const float ox = p0 * cx + p1 * cy + p2 * cz + p3, oy = p4 * cx + p5 * cy + p6 * cz + p7, oz = p8 * cx + p9 * cy + p10 * cz + p11;
    out[i * 3 + 0] = ox;
    out[i * 3 + 1] = oy;
    out[i * 3 + 2] = oz;
    // ---- the check ----
    const f4 r0 = reload4(p), r1 = reload4(p + 4), r2 = reload4(p + 8);
    const f4 k0 = reload4(k), k1 = reload4(k + 4), k2 = reload4(k + 8);
    float z2;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(z2) : "v"(depth + (long long)f * H * W + pix) : "memory");
    unsigned fl = 0;
    if (p0 != r0.x || p1 != r0.y || p2 != r0.z || p3 != r0.w) fl |= 1;
    if (p4 != r1.x || p5 != r1.y || p6 != r1.z || p7 != r1.w) fl |= 2;
    if (p8 != r2.x || p9 != r2.y || p10 != r2.z || p11 != r2.w) fl |= 4;
    if (a != k0.x || b != k0.y || c != k0.z) fl |= 16;
    if (d != k1.x || e != k1.y || g != k1.z) fl |= 32;
    if (h != k2.x || l != k2.y || m != k2.z) fl |= 64;
    if (z != z2) fl |= 128;
    // 256: the value the kernel COMPUTED differs from the same expression on the re-read inputs; 512: what memory holds behind the
    // store differs from what was computed
    {
        const float ox2 = r0.x * cx + r0.y * cy + r0.z * cz + r0.w, oy2 = r1.x * cx + r1.y * cy + r1.z * cz + r1.w,
                    oz2 = r2.x * cx + r2.y * cy + r2.z * cz + r2.w;
        if (ox != ox2 || oy != oy2 || oz != oz2) fl |= 256;
        float b0, b1, b2;
        asm volatile("s_waitcnt vmcnt(0)\n\tglobal_load_dword %0, %3, off sc0 sc1\n\tglobal_load_dword %1, %3, off offset:4 sc0 sc1\n\t"
                     "global_load_dword %2, %3, off offset:8 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(b0), "=&v"(b1), "=&v"(b2) : "v"(out + i * 3) : "memory");
        if (b0 != ox || b1 != oy || b2 != oz) fl |= 512;
    }
    if (fl) {
        const unsigned slot = atomicAdd(nlog, 1u);
        if (slot < cap) {
            Rec& R = log[slot];
            R.launch = launch, R.iter = 0, R.gid = i, R.row = 0;
            R.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            R.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            R.flags = fl;
            if (fl & 1) R.a[0] = p0, R.a[1] = p1, R.a[2] = p2, R.a[3] = p3, R.b[0] = r0.x, R.b[1] = r0.y, R.b[2] = r0.z, R.b[3] = r0.w;
            else if (fl & 2) R.a[0] = p4, R.a[1] = p5, R.a[2] = p6, R.a[3] = p7, R.b[0] = r1.x, R.b[1] = r1.y, R.b[2] = r1.z, R.b[3] = r1.w;
            else if (fl & 4) R.a[0] = p8, R.a[1] = p9, R.a[2] = p10, R.a[3] = p11, R.b[0] = r2.x, R.b[1] = r2.y, R.b[2] = r2.z, R.b[3] = r2.w;
            else R.a[0] = a, R.a[1] = e, R.a[2] = m, R.a[3] = z, R.b[0] = k0.x, R.b[1] = k1.y, R.b[2] = k2.z, R.b[3] = z2;
            R.c[0] = ox, R.c[1] = oy, R.c[2] = oz, R.c[3] = 0.f;
            R.t = __builtin_readcyclecounter();
            R.addr = (unsigned long long)p;
        }
    }
}
extern "C" int race_victim_full_launch(void* stream, const float* depth, const float* K, const float* P, float* out, int F, int H,
                                       int W, int ratio, unsigned seed, unsigned launch, void* log, unsigned* nlog, unsigned cap) {
    const int spf = (H * W) / ratio;
    const int mode = (int)(launch >> 30);  // top two bits of the launch number select the variant (0 checked, 1 drained, 2 nops)
    if (mode == 1)
        hipLaunchKernelGGL(victim_full_kernel<1>, dim3((F * spf + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth, K, P, out, F, H,
                           W, ratio, seed, spf, launch, (Rec*)log, nlog, cap);
    else if (mode == 2)
        hipLaunchKernelGGL(victim_full_kernel<2>, dim3((F * spf + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth, K, P, out, F, H,
                           W, ratio, seed, spf, launch, (Rec*)log, nlog, cap);
    else
    hipLaunchKernelGGL(victim_full_kernel<0>, dim3((F * spf + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth, K, P, out, F, H, W,
                       ratio, seed, spf, launch, (Rec*)log, nlog, cap);
    return (int)hipGetLastError();
}

// Counted waits under test: the pointmap kernel's seven loads (3 x dwordx3 of K, the depth gather, 3 x dwordx4 of the pose) in
// flight together, destinations pre-filled with a sentinel (777.0), every destination copied IMMEDIATELY behind the counted
// s_waitcnt that releases it (as the compiler's code consumes it), and compared with the same register after vmcnt(0) + delay.
// acc[j] != 0: load j's registers were not final when its counted wait let the wave through.
__global__ void victim_cnt_kernel(const float* __restrict__ depth, const float* __restrict__ K, const float* __restrict__ P,
                                  float* __restrict__ out, int F, int H, int W, int ratio, unsigned seed, int spf, int iters,
                                  unsigned launch, Rec* __restrict__ log, unsigned* __restrict__ nlog, unsigned cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * spf) return;
    const int f = i / spf, j = i % spf;
    int pix = j * ratio + (int)(hash_u32(seed ^ (unsigned)j * 2654435761u) % (unsigned)ratio);
    if (pix >= H * W) pix = H * W - 1;
    const unsigned long long ka = (unsigned long long)(K + f * 16), da = (unsigned long long)(depth + (long long)f * H * W + pix),
                             pa = (unsigned long long)(P + f * 16);
    const unsigned klo = (unsigned)ka, khi = (unsigned)(ka >> 32), dlo = (unsigned)da, dhi = (unsigned)(da >> 32), plo = (unsigned)pa,
                   phi = (unsigned)(pa >> 32);
    float sum = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned acc[7];
        float e0, l0, e1, l1;
            asm volatile(
                "v_mov_b32 v28, %11\n\t"
                "v_mov_b32 v29, %12\n\t"
                "v_mov_b32 v30, %13\n\t"
                "v_mov_b32 v31, %14\n\t"
                "v_mov_b32 v32, %15\n\t"
                "v_mov_b32 v33, %16\n\t"
                "s_mov_b32 s40, 0x44424000\n\t"
                "v_mov_b32 v0, s40\n\t"
                "v_mov_b32 v1, s40\n\t"
                "v_mov_b32 v2, s40\n\t"
                "v_mov_b32 v4, s40\n\t"
                "v_mov_b32 v5, s40\n\t"
                "v_mov_b32 v6, s40\n\t"
                "v_mov_b32 v8, s40\n\t"
                "v_mov_b32 v9, s40\n\t"
                "v_mov_b32 v10, s40\n\t"
                "v_mov_b32 v12, s40\n\t"
                "v_mov_b32 v16, s40\n\t"
                "v_mov_b32 v17, s40\n\t"
                "v_mov_b32 v18, s40\n\t"
                "v_mov_b32 v19, s40\n\t"
                "v_mov_b32 v20, s40\n\t"
                "v_mov_b32 v21, s40\n\t"
                "v_mov_b32 v22, s40\n\t"
                "v_mov_b32 v23, s40\n\t"
                "v_mov_b32 v24, s40\n\t"
                "v_mov_b32 v25, s40\n\t"
                "v_mov_b32 v26, s40\n\t"
                "v_mov_b32 v27, s40\n\t"
                "s_nop 4\n\t"
                "global_load_dwordx3 v[0:2], v[28:29], off\n\t"
                "global_load_dwordx3 v[4:6], v[28:29], off offset:16\n\t"
                "global_load_dwordx3 v[8:10], v[28:29], off offset:32\n\t"
                "global_load_dword v12, v[30:31], off\n\t"
                "global_load_dwordx4 v[16:19], v[32:33], off offset:16\n\t"
                "global_load_dwordx4 v[20:23], v[32:33], off\n\t"
                "global_load_dwordx4 v[24:27], v[32:33], off offset:32\n\t"
                "s_waitcnt vmcnt(6)\n\t"
                "v_mov_b32 v40, v0\n\t"
                "v_mov_b32 v41, v1\n\t"
                "v_mov_b32 v42, v2\n\t"
                "s_waitcnt vmcnt(5)\n\t"
                "v_mov_b32 v43, v4\n\t"
                "v_mov_b32 v44, v5\n\t"
                "v_mov_b32 v45, v6\n\t"
                "s_waitcnt vmcnt(4)\n\t"
                "v_mov_b32 v46, v8\n\t"
                "v_mov_b32 v47, v9\n\t"
                "v_mov_b32 v48, v10\n\t"
                "s_waitcnt vmcnt(3)\n\t"
                "v_mov_b32 v49, v12\n\t"
                "s_waitcnt vmcnt(2)\n\t"
                "v_mov_b32 v50, v16\n\t"
                "v_mov_b32 v51, v17\n\t"
                "v_mov_b32 v52, v18\n\t"
                "v_mov_b32 v53, v19\n\t"
                "s_waitcnt vmcnt(1)\n\t"
                "v_mov_b32 v54, v20\n\t"
                "v_mov_b32 v55, v21\n\t"
                "v_mov_b32 v56, v22\n\t"
                "v_mov_b32 v57, v23\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "v_mov_b32 v58, v24\n\t"
                "v_mov_b32 v59, v25\n\t"
                "v_mov_b32 v60, v26\n\t"
                "v_mov_b32 v61, v27\n\t"
                "s_nop 7\n\t"
                "s_nop 7\n\t"
                "s_nop 7\n\t"
                "s_nop 7\n\t"
                "s_nop 7\n\t"
                "s_nop 7\n\t"
                "v_mov_b32 v62, 0\n\t"
                "v_xor_b32 v70, v40, v0\n\t"
                "v_or_b32 v62, v62, v70\n\t"
                "v_xor_b32 v70, v41, v1\n\t"
                "v_or_b32 v62, v62, v70\n\t"
                "v_xor_b32 v70, v42, v2\n\t"
                "v_or_b32 v62, v62, v70\n\t"
                "v_mov_b32 v63, 0\n\t"
                "v_xor_b32 v70, v43, v4\n\t"
                "v_or_b32 v63, v63, v70\n\t"
                "v_xor_b32 v70, v44, v5\n\t"
                "v_or_b32 v63, v63, v70\n\t"
                "v_xor_b32 v70, v45, v6\n\t"
                "v_or_b32 v63, v63, v70\n\t"
                "v_mov_b32 v64, 0\n\t"
                "v_xor_b32 v70, v46, v8\n\t"
                "v_or_b32 v64, v64, v70\n\t"
                "v_xor_b32 v70, v47, v9\n\t"
                "v_or_b32 v64, v64, v70\n\t"
                "v_xor_b32 v70, v48, v10\n\t"
                "v_or_b32 v64, v64, v70\n\t"
                "v_mov_b32 v65, 0\n\t"
                "v_xor_b32 v70, v49, v12\n\t"
                "v_or_b32 v65, v65, v70\n\t"
                "v_mov_b32 v66, 0\n\t"
                "v_xor_b32 v70, v50, v16\n\t"
                "v_or_b32 v66, v66, v70\n\t"
                "v_xor_b32 v70, v51, v17\n\t"
                "v_or_b32 v66, v66, v70\n\t"
                "v_xor_b32 v70, v52, v18\n\t"
                "v_or_b32 v66, v66, v70\n\t"
                "v_xor_b32 v70, v53, v19\n\t"
                "v_or_b32 v66, v66, v70\n\t"
                "v_mov_b32 v67, 0\n\t"
                "v_xor_b32 v70, v54, v20\n\t"
                "v_or_b32 v67, v67, v70\n\t"
                "v_xor_b32 v70, v55, v21\n\t"
                "v_or_b32 v67, v67, v70\n\t"
                "v_xor_b32 v70, v56, v22\n\t"
                "v_or_b32 v67, v67, v70\n\t"
                "v_xor_b32 v70, v57, v23\n\t"
                "v_or_b32 v67, v67, v70\n\t"
                "v_mov_b32 v68, 0\n\t"
                "v_xor_b32 v70, v58, v24\n\t"
                "v_or_b32 v68, v68, v70\n\t"
                "v_xor_b32 v70, v59, v25\n\t"
                "v_or_b32 v68, v68, v70\n\t"
                "v_xor_b32 v70, v60, v26\n\t"
                "v_or_b32 v68, v68, v70\n\t"
                "v_xor_b32 v70, v61, v27\n\t"
                "v_or_b32 v68, v68, v70\n\t"
                "v_mov_b32 %0, v62\n\t"
                "v_mov_b32 %1, v63\n\t"
                "v_mov_b32 %2, v64\n\t"
                "v_mov_b32 %3, v65\n\t"
                "v_mov_b32 %4, v66\n\t"
                "v_mov_b32 %5, v67\n\t"
                "v_mov_b32 %6, v68\n\t"
                "v_mov_b32 %7, v54\n\t"
                "v_mov_b32 %8, v20\n\t"
                "v_mov_b32 %9, v51\n\t"
                "v_mov_b32 %10, v17\n\t"
                : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]), "=&v"(acc[6]), "=&v"(e0), "=&v"(l0), "=&v"(e1), "=&v"(l1)
                : "v"(klo), "v"(khi), "v"(dlo), "v"(dhi), "v"(plo), "v"(phi)
                : "v0", "v1", "v2", "v4", "v5", "v6", "v8", "v9", "v10", "v12", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v70", "s40", "memory");
        unsigned fl = 0;
        for (int q = 0; q < 7; ++q) fl |= acc[q] ? 1u << q : 0u;
        if (fl) {
            const unsigned slot = atomicAdd(nlog, 1u);
            if (slot < cap) {
                Rec& R = log[slot];
                R.launch = launch, R.iter = it, R.gid = i, R.row = 0;
                R.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
                R.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                R.flags = fl;
                R.a[0] = e0, R.a[1] = l0, R.a[2] = e1, R.a[3] = l1;
                for (int c = 0; c < 4; ++c) R.b[c] = __uint_as_float(acc[c]), R.c[c] = c < 3 ? __uint_as_float(acc[4 + c]) : 0.f;
                R.t = __builtin_readcyclecounter();
                R.addr = pa;
            }
        }
        sum += l0 + l1;
    }
    out[i * 3 + 0] = sum;
    out[i * 3 + 1] = 0.f;
    out[i * 3 + 2] = 0.f;
}
extern "C" int race_victim_cnt_launch(void* stream, const float* depth, const float* K, const float* P, float* out, int F, int H,
                                      int W, int ratio, unsigned seed, int iters, unsigned launch, void* log, unsigned* nlog,
                                      unsigned cap) {
    const int spf = (H * W) / ratio;
    hipLaunchKernelGGL(victim_cnt_kernel, dim3((F * spf + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth, K, P, out, F, H, W,
                       ratio, seed, spf, iters, launch, (Rec*)log, nlog, cap);
    return (int)hipGetLastError();
}

// Register canary: a wave parks known values in v0..v95, sleeps (spin x s_sleep 2), then dumps the registers to memory
// (dump[r][thread]).  Nothing in this kernel writes them in between: a changed value is a write from OUTSIDE the wave - another
// wave on the same SIMD addressing registers beyond its own allocation.
__global__ void canary_kernel(unsigned* __restrict__ dump, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long a = (unsigned long long)(dump + i);
    const unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
    const int stride = n * 4;
    asm volatile(
        "v_mov_b32 v96, %0\n\t"
        "v_mov_b32 v97, %1\n\t"
        "s_mov_b32 s40, %2\n\t"
        "v_mov_b32 v0, 0x5a00005a\n\t"
        "v_mov_b32 v1, 0x5a00015a\n\t"
        "v_mov_b32 v2, 0x5a00025a\n\t"
        "v_mov_b32 v3, 0x5a00035a\n\t"
        "v_mov_b32 v4, 0x5a00045a\n\t"
        "v_mov_b32 v5, 0x5a00055a\n\t"
        "v_mov_b32 v6, 0x5a00065a\n\t"
        "v_mov_b32 v7, 0x5a00075a\n\t"
        "v_mov_b32 v8, 0x5a00085a\n\t"
        "v_mov_b32 v9, 0x5a00095a\n\t"
        "v_mov_b32 v10, 0x5a000a5a\n\t"
        "v_mov_b32 v11, 0x5a000b5a\n\t"
        "v_mov_b32 v12, 0x5a000c5a\n\t"
        "v_mov_b32 v13, 0x5a000d5a\n\t"
        "v_mov_b32 v14, 0x5a000e5a\n\t"
        "v_mov_b32 v15, 0x5a000f5a\n\t"
        "v_mov_b32 v16, 0x5a00105a\n\t"
        "v_mov_b32 v17, 0x5a00115a\n\t"
        "v_mov_b32 v18, 0x5a00125a\n\t"
        "v_mov_b32 v19, 0x5a00135a\n\t"
        "v_mov_b32 v20, 0x5a00145a\n\t"
        "v_mov_b32 v21, 0x5a00155a\n\t"
        "v_mov_b32 v22, 0x5a00165a\n\t"
        "v_mov_b32 v23, 0x5a00175a\n\t"
        "v_mov_b32 v24, 0x5a00185a\n\t"
        "v_mov_b32 v25, 0x5a00195a\n\t"
        "v_mov_b32 v26, 0x5a001a5a\n\t"
        "v_mov_b32 v27, 0x5a001b5a\n\t"
        "v_mov_b32 v28, 0x5a001c5a\n\t"
        "v_mov_b32 v29, 0x5a001d5a\n\t"
        "v_mov_b32 v30, 0x5a001e5a\n\t"
        "v_mov_b32 v31, 0x5a001f5a\n\t"
        "v_mov_b32 v32, 0x5a00205a\n\t"
        "v_mov_b32 v33, 0x5a00215a\n\t"
        "v_mov_b32 v34, 0x5a00225a\n\t"
        "v_mov_b32 v35, 0x5a00235a\n\t"
        "v_mov_b32 v36, 0x5a00245a\n\t"
        "v_mov_b32 v37, 0x5a00255a\n\t"
        "v_mov_b32 v38, 0x5a00265a\n\t"
        "v_mov_b32 v39, 0x5a00275a\n\t"
        "v_mov_b32 v40, 0x5a00285a\n\t"
        "v_mov_b32 v41, 0x5a00295a\n\t"
        "v_mov_b32 v42, 0x5a002a5a\n\t"
        "v_mov_b32 v43, 0x5a002b5a\n\t"
        "v_mov_b32 v44, 0x5a002c5a\n\t"
        "v_mov_b32 v45, 0x5a002d5a\n\t"
        "v_mov_b32 v46, 0x5a002e5a\n\t"
        "v_mov_b32 v47, 0x5a002f5a\n\t"
        "v_mov_b32 v48, 0x5a00305a\n\t"
        "v_mov_b32 v49, 0x5a00315a\n\t"
        "v_mov_b32 v50, 0x5a00325a\n\t"
        "v_mov_b32 v51, 0x5a00335a\n\t"
        "v_mov_b32 v52, 0x5a00345a\n\t"
        "v_mov_b32 v53, 0x5a00355a\n\t"
        "v_mov_b32 v54, 0x5a00365a\n\t"
        "v_mov_b32 v55, 0x5a00375a\n\t"
        "v_mov_b32 v56, 0x5a00385a\n\t"
        "v_mov_b32 v57, 0x5a00395a\n\t"
        "v_mov_b32 v58, 0x5a003a5a\n\t"
        "v_mov_b32 v59, 0x5a003b5a\n\t"
        "v_mov_b32 v60, 0x5a003c5a\n\t"
        "v_mov_b32 v61, 0x5a003d5a\n\t"
        "v_mov_b32 v62, 0x5a003e5a\n\t"
        "v_mov_b32 v63, 0x5a003f5a\n\t"
        "v_mov_b32 v64, 0x5a00405a\n\t"
        "v_mov_b32 v65, 0x5a00415a\n\t"
        "v_mov_b32 v66, 0x5a00425a\n\t"
        "v_mov_b32 v67, 0x5a00435a\n\t"
        "v_mov_b32 v68, 0x5a00445a\n\t"
        "v_mov_b32 v69, 0x5a00455a\n\t"
        "v_mov_b32 v70, 0x5a00465a\n\t"
        "v_mov_b32 v71, 0x5a00475a\n\t"
        "v_mov_b32 v72, 0x5a00485a\n\t"
        "v_mov_b32 v73, 0x5a00495a\n\t"
        "v_mov_b32 v74, 0x5a004a5a\n\t"
        "v_mov_b32 v75, 0x5a004b5a\n\t"
        "v_mov_b32 v76, 0x5a004c5a\n\t"
        "v_mov_b32 v77, 0x5a004d5a\n\t"
        "v_mov_b32 v78, 0x5a004e5a\n\t"
        "v_mov_b32 v79, 0x5a004f5a\n\t"
        "v_mov_b32 v80, 0x5a00505a\n\t"
        "v_mov_b32 v81, 0x5a00515a\n\t"
        "v_mov_b32 v82, 0x5a00525a\n\t"
        "v_mov_b32 v83, 0x5a00535a\n\t"
        "v_mov_b32 v84, 0x5a00545a\n\t"
        "v_mov_b32 v85, 0x5a00555a\n\t"
        "v_mov_b32 v86, 0x5a00565a\n\t"
        "v_mov_b32 v87, 0x5a00575a\n\t"
        "v_mov_b32 v88, 0x5a00585a\n\t"
        "v_mov_b32 v89, 0x5a00595a\n\t"
        "v_mov_b32 v90, 0x5a005a5a\n\t"
        "v_mov_b32 v91, 0x5a005b5a\n\t"
        "v_mov_b32 v92, 0x5a005c5a\n\t"
        "v_mov_b32 v93, 0x5a005d5a\n\t"
        "v_mov_b32 v94, 0x5a005e5a\n\t"
        "v_mov_b32 v95, 0x5a005f5a\n\t"
        "1:\n\t"
        "s_sleep 2\n\t"
        "s_sub_u32 s40, s40, 1\n\t"
        "s_cmp_lg_u32 s40, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b32 s41, %3\n\t"
        "global_store_dword v[96:97], v0, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v1, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v2, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v3, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v4, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v5, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v6, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v7, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v8, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v9, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v10, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v11, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v12, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v13, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v14, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v15, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v16, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v17, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v18, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v19, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v20, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v21, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v22, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v23, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v24, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v25, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v26, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v27, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v28, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v29, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v30, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v31, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v32, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v33, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v34, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v35, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v36, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v37, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v38, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v39, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v40, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v41, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v42, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v43, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v44, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v45, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v46, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v47, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v48, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v49, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v50, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v51, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v52, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v53, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v54, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v55, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v56, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v57, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v58, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v59, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v60, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v61, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v62, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v63, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v64, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v65, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v66, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v67, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v68, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v69, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v70, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v71, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v72, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v73, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v74, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v75, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v76, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v77, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v78, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v79, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v80, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v81, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v82, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v83, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v84, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v85, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v86, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v87, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v88, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v89, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v90, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v91, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v92, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v93, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v94, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "global_store_dword v[96:97], v95, off\n\t"
        "v_add_co_u32 v96, vcc, s41, v96\n\t"
        "v_addc_co_u32 v97, vcc, 0, v97, vcc\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        :
        : "v"(lo), "v"(hi), "s"(spin), "s"(stride)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "s40", "s41", "vcc", "memory");
}
extern "C" int race_canary_launch(void* stream, unsigned* dump, int n, int spin) {
    hipLaunchKernelGGL(canary_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dump, n, spin);
    return (int)hipGetLastError();
}

// Synthetic aggressors: chip-filling kernels that do nothing but ONE instruction class, to run on a second stream beside the
// victim.  kind: 0 plain VALU (v_fma_f32), 1 v_permlane32_swap, 2 v_permlane16_swap, 3 MFMA 32x32x16 bf16, 4 v_pk_fma_f32,
// 5 ds_read_b64_tr_b16, 6 plain VALU at s_setprio 3, 7 v_exp_f32, 8 DPP row_shr, 9 v_pk_mul_f32 with op_sel:[0,1], 10 MFMA 16x16x32 bf16,
// 11 global_load_lds_dwordx4 (LDS-DMA), 12 v_cvt_pk_bf16_f32
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void __launch_bounds__(256) aggressor_kernel(float* __restrict__ sink, const float* __restrict__ src, int iters) {
    __shared__ float lds[4096];
    const int t = threadIdx.x;
    float a = 1.0f + t * 1e-3f, b = 0.5f + t * 1e-4f;
    f2 p = {a, b}, q = {b, a};
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    f16v acc = {0};
    f4 acc4 = {0, 0, 0, 0};
    bf8 va, vb;
    for (int k = 0; k < 8; ++k) va[k] = (__bf16)(0.01f * (t + k)), vb[k] = (__bf16)(0.02f * (t - k));
    lds[t] = a;
    lds[t + 256] = b;
    __syncthreads();
    if (KIND == 6) asm volatile("s_setprio 3");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0 || KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
            if (KIND == 1) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
            if (KIND == 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
            if (KIND == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
            if (KIND == 10) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc4, 0, 0, 0);
            if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
            if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p) : "v"(q));
            if (KIND == 5) {
                f2 r;
                asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((t & 63) * 8) : "memory");
                p += r;
            }
            if (KIND == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
            if (KIND == 8) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
            if (KIND == 11) {
                asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" : : "v"(src + (t & 63) * 4), "s"(0) : "memory", "m0");
            }
            if (KIND == 12) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ua) : "v"(a), "v"(b));
        }
    }
    float s = a + b + p.x + p.y + __uint_as_float(ua) + __uint_as_float(ub) + acc[0] + acc[5] + acc4[1] + lds[(t * 7) & 511];
    if (s == 123.456f) sink[t] = s;
}
extern "C" int race_aggr_launch(void* stream, int kind, int blocks, int iters, float* sink, const float* src) {
#define AG(K) case K: hipLaunchKernelGGL(aggressor_kernel<K>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, src, iters); break;
    switch (kind) { AG(0) AG(1) AG(2) AG(3) AG(4) AG(5) AG(6) AG(7) AG(8) AG(9) AG(10) AG(11) AG(12) default: return -1; }
    return (int)hipGetLastError();
}

// ---- second round: which MFMA forms disturb which VALU forms -------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
// aggressor2 kind: 0 16x16x32 bf16 (dependent chain), 1 16x16x32 bf16 (4 independent accumulators), 2 16x16x32 f16, 3 16x16x16 bf16_1k,
// 4 16x16x4 f32, 5 32x32x2 f32, 6 16x16x32 fp8, 7 32x32x16 f16, 8 32x32x16 bf16, 9 4x4x4 f16 (4 f16 per lane), 10 32x32x8 bf16_1k
template <int KIND>
__global__ void __launch_bounds__(256) aggressor2_kernel(float* __restrict__ sink, int iters) {
    const int t = threadIdx.x;
    bf8 va, vb;
    h8 ha, hb;
    s4v sa, sb;
    for (int k = 0; k < 8; ++k) va[k] = (__bf16)(0.01f * (t + k)), vb[k] = (__bf16)(0.02f * (t - k)), ha[k] = (_Float16)(0.01f * k), hb[k] = (_Float16)(0.5f);
    for (int k = 0; k < 4; ++k) sa[k] = (short)(0x3f80 + t + k), sb[k] = (short)(0x3f00 + k);
    const long la = 0x3839383938393839L + t, lb = 0x4040404040404040L;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 qa = {(_Float16)1.f, (_Float16)(0.01f * t), (_Float16)0.5f, (_Float16)2.f}, qb = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)3.f};
    const float fa = 1.f + t, fb = 0.5f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f16v w = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, c0, 0, 0, 0);
            if (KIND == 1) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, c3, 0, 0, 0);
            }
            if (KIND == 2) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
            if (KIND == 3) c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c0, 0, 0, 0);
            if (KIND == 4) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
            if (KIND == 5) w = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, w, 0, 0, 0);
            if (KIND == 6) c0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c0, 0, 0, 0);
            if (KIND == 7) w = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, w, 0, 0, 0);
            if (KIND == 8) w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, w, 0, 0, 0);
            if (KIND == 9) c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(qa, qb, c0, 0, 0, 0);
            if (KIND == 10) w = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(sa, sb, w, 0, 0, 0);
        }
    }
    const float s = c0[0] + c1[1] + c2[2] + c3[3] + w[0] + w[7];
    if (s == 123.456f) sink[t] = s;
}

extern "C" int race_aggr2_launch(void* stream, int kind, int blocks, int iters, float* sink) {
#define AG2(K) case K: hipLaunchKernelGGL(aggressor2_kernel<K>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters); break;
    switch (kind) { AG2(0) AG2(1) AG2(2) AG2(3) AG2(4) AG2(5) AG2(6) AG2(7) AG2(8) AG2(9) AG2(10) default: return -1; }
    return (int)hipGetLastError();
}

// VALU forms under test (victim side).  Every lane repeats ONE instruction on fixed, lane-dependent inputs; mode 0 writes the two
// result dwords (reference launch, nothing else running), mode 1 counts the iterations whose result differs from that reference.
// form: 0 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]   1 v_pk_mul_f32   2 v_pk_fma_f32   3 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]
//       4 v_pk_mul_f32 op_sel_hi:[1,0]   5 v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]   6 v_pk_mov_b32 op_sel:[1,0]   7 v_mul_f32
//       8 v_fma_f32   9 v_pk_fma_f16   10 v_fma_f64   11 v_mad_u64_u32   12 v_lshl_add_u64   13 v_cvt_pk_bf16_f32   14 v_pk_add_f32
//       15 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]   16 v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0]   17 v_pk_fma_f32 op_sel_hi:[1,0,1]
//       18 v_pk_mul_f16 op_sel:[0,1] op_sel_hi:[1,0]   19 v_add_f32_dpp row_shr:1   20 v_permlane32_swap   21 v_mul_f64
template <int FORM>
__global__ void __launch_bounds__(256) valu_victim_kernel(unsigned* __restrict__ ref, unsigned* __restrict__ cnt, int n, int iters, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const f2 a = {1.0f + 0.001f * (i % 977), 2.0f + 0.003f * (i % 613)}, b = {0.5f + 0.002f * (i % 331), 1.5f + 0.001f * (i % 127)},
             c = {0.25f + 0.004f * (i % 89), 3.0f - 0.002f * (i % 53)};
    const u2 r0 = {ref[2 * i], ref[2 * i + 1]};
    unsigned bad = 0, badlo = 0;
    u2 d = {0, 0};
    for (int it = 0; it < iters; ++it) {
        f2 x = a;
        d = (u2){0, 0};
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(x), "v"(b), "v"(c));
        if (FORM == 3) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 4) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 6) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 7) asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(d.x) : "v"(x.x), "v"(b.x));
        if (FORM == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(d.x) : "v"(x.x), "v"(b.x), "v"(c.x));
        if (FORM == 9) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=&v"(d.x) : "v"(x.x), "v"(b.x), "v"(c.x));
        if (FORM == 10) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=&v"(d) : "v"(x), "v"(b), "v"(c));
        if (FORM == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(d) : "v"(x.x), "v"(b.x), "v"(c) : "vcc");
        if (FORM == 12) asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=&v"(d.x) : "v"(x.x), "v"(b.x));
        if (FORM == 14) asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 15) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 16) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 17) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(d) : "v"(x), "v"(b), "v"(c));
        if (FORM == 18) asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d.x) : "v"(x.x), "v"(b.x));
        if (FORM == 19) asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(d.x) : "v"(x.x), "v"(b.x));
        if (FORM == 20) {
            unsigned p = __float_as_uint(x.x), q = __float_as_uint(b.x);
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
            d = (u2){p, q};
        }
        if (FORM == 21) asm volatile("v_mul_f64 %0, %1, %2" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 22) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(d) : "v"(x), "v"(b), "v"(c));
        if (FORM == 23) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(d) : "v"(x), "v"(b), "v"(c));
        if (FORM == 24) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=&v"(d) : "v"(x), "v"(b), "v"(c));
        if (FORM == 25) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 26) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 27) asm volatile("v_pk_mul_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(x), "v"(b));
        if (FORM == 28) asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(d) : "v"(x), "v"(b));
        if (mode) {
            bad += (d.x != r0.x || d.y != r0.y) ? 1u : 0u;
            badlo += (d.x != r0.x) ? 1u : 0u;
        }
    }
    if (mode == 0) {
        ref[2 * i] = d.x;
        ref[2 * i + 1] = d.y;
    } else {
        cnt[2 * i] = bad;
        cnt[2 * i + 1] = badlo;
    }
}
extern "C" int race_valu_launch(void* stream, int form, unsigned* ref, unsigned* cnt, int n, int iters, int mode) {
#define VV(K) case K: hipLaunchKernelGGL(valu_victim_kernel<K>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ref, cnt, n, iters, mode); break;
    switch (form) { VV(0) VV(1) VV(2) VV(3) VV(4) VV(5) VV(6) VV(7) VV(8) VV(9) VV(10) VV(11) VV(12) VV(13) VV(14) VV(15) VV(16) VV(17) VV(18) VV(19) VV(20) VV(21) VV(22) VV(23) VV(24) VV(25) VV(26) VV(27) VV(28) default: return -1; }
    return (int)hipGetLastError();
}
