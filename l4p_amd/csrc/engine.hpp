// Engine state + launcher prototypes shared by the translation units of libl4p_hip.so.
#pragma once
#include <string>
#include <unordered_map>

#include "gemm.hpp"

struct Weight {
    const void* ptr;
    long long numel;
};

struct l4p_engine {
    int device = 0;
    int dtype = L4P_BF16;
    std::unordered_map<std::string, Weight> w;
    l4p_encoder_cfg enc{};
    int enc_tokens = 0;
    bool enc_set = false;

    const void* find(const std::string& key) const {
        auto it = w.find(key);
        return it == w.end() ? nullptr : it->second.ptr;
    }
};

int launch_layernorm(int dtype, const float* x, const float* gamma, const float* beta, float eps, void* out_T,
                     float* out_f32, int M, int C, hipStream_t stream);
int launch_cast(int dtype, const float* x, void* y, long long n, hipStream_t stream);
int launch_patch_gather(int dtype, const float* rgb, void* out, int B, int Cin, int T, int H, int W, int pt, int ph,
                        int pw, int Kp, hipStream_t stream);
int launch_attention(int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                     hipStream_t stream);
int launch_attention64(int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                       hipStream_t stream);  // attention64.hip
int launch_upsample(int dtype, const void* x, void* y, int B, int Ti, int Hi, int Wi, int To, int Ho, int Wo, int C,
                    int align, hipStream_t stream);
int launch_head_out(int dtype, const void* x, const float* w, const float* bias, float* y, long long vox_per_b, int B,
                    int C, int Cout, int post_exp, hipStream_t stream);
int launch_affine_solve(const float* pred, const float* tgt, long long n, int inverse, double* scratch, float* sol,
                        hipStream_t stream);
int launch_affine_apply(const float* x, float* y, long long n, int inverse, const float* sol, hipStream_t stream);
int launch_rays_to_pose_rot(const float* rays, const float* R, float* out, int B, int T, int h, int w, hipStream_t stream);
int launch_rays_to_pose(const float* rays, const float* K, float* out, int B, int T, int h, int w, int H, int W,
                        hipStream_t stream);
