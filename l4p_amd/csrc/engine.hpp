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
int launch_attention(int dtype, const void* qk, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                     hipStream_t stream);
