// Optional per-kernel-class timing with HIP events recorded on the launch stream (bench.py's live
// roofline measurement).  Disabled by default: a ProfScope is then two predictable branches.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

enum ProfClass {
    PROF_GEMM = 0,      // gemm8p / gemm_kernel<..., MODE 0>: linears, 1x1x1 convs, ConvTranspose-as-GEMM (>= 1024 rows, one weight matrix)
    PROF_CONV3D,        // gemm_kernel<..., MODE 1>: implicit-GEMM 3x3x3 conv
    PROF_ATTENTION,     // attn_kernel
    PROF_LAYERNORM,
    PROF_ELEMENTWISE,   // cast / patch gather / upsample / head_out / alignment / pose
    PROF_TRACK,         // tracker-specific small kernels
    PROF_PREP,          // clip preparation (preprocess.hip): Pillow resample passes, resize + normalise
    PROF_GEMM_SMALL,    // dense products that are latency- or HBM-bound by construction: fewer than 1024 rows (the tracker's token-side
                        // projections, the coarsest DPT levels) and the tracker's folded cross-attention products (row-grouped weights,
                        // l4p_t2i_context, l4p_i2t_delta) - kept out of PROF_GEMM, whose time / FLOPs price the MFMA-bound kernels
    PROF_NUM
};

// Dispatch knobs (l4p_set_knob / l4p_get_knob, include/l4p_hip.h): environment default read once, run-time setter for the tests.
enum Knob { KNOB_CONV_HALO = 0, KNOB_GEMM_4W, KNOB_MASKDOT_MFMA, KNOB_CONV_UPS, KNOB_LN_TRACKS, KNOB_LN_ROWS16, KNOB_ATTN64, KNOB_GEMM_SKINNY, KNOB_READOUT_WIDE, KNOB_TRACK_DEEP, KNOB_PROBE_KERNELS, KNOB_NUM };
int knob(int id);

void prof_begin(int cls, hipStream_t stream, const char* tag = nullptr);
void prof_end(int cls, hipStream_t stream);
extern bool g_prof_on;

struct ProfScope {
    int cls;
    hipStream_t s;
    ProfScope(int c, hipStream_t st) : cls(c), s(st) {
        if (g_prof_on) prof_begin(cls, s);
    }
    // tagged form: the printf-style tag (e.g. the GEMM shape) keys the per-tag table of l4p_prof_detail()
    ProfScope(int c, hipStream_t st, const char* fmt, ...) __attribute__((format(printf, 4, 5))) : cls(c), s(st) {
        if (!g_prof_on) return;
        char tag[64];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(tag, sizeof tag, fmt, ap);
        va_end(ap);
        prof_begin(cls, s, tag);
    }
    ~ProfScope() {
        if (g_prof_on) prof_end(cls, s);
    }
};
