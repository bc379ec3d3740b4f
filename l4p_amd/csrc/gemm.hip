#include "gemm.hpp"
int launch_gemm_bf16(int mode, const GemmParams& p, hipStream_t stream);
int launch_gemm_f16(int mode, const GemmParams& p, hipStream_t stream);
int launch_gemm_f32(int mode, const GemmParams& p, hipStream_t stream);

int launch_gemm(int dtype, int mode, const GemmParams& p, hipStream_t stream) {
    if (!dtype_ok(dtype)) { l4p_set_error("gemm: unknown dtype %d", dtype); return L4P_E_INVALID; }
    const int es = esize_of(dtype);
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) { l4p_set_error("gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K); return L4P_E_INVALID; }
    if (p.N % 8) { l4p_set_error("gemm: N=%d must be a multiple of 8", p.N); return L4P_E_INVALID; }
    if ((p.K * es) % 16 || (p.ldw * es) % 16) { l4p_set_error("gemm: K/ldw not 16-byte aligned"); return L4P_E_INVALID; }
    if (mode == 0 && (p.lda * es) % 16) { l4p_set_error("gemm: lda not 16-byte aligned"); return L4P_E_INVALID; }
    if (mode == 1 && (p.Cin % (128 / es) || p.K != 27 * p.Cin)) { l4p_set_error("conv3d: Cin=%d must be a multiple of %d and K=27*Cin", p.Cin, 128 / es); return L4P_E_INVALID; }
    if (p.ups_hi > 0 && (mode != 1 || !is16(dtype) || p.ups_wi <= 0 || p.To != p.Ti)) {
        l4p_set_error("conv3d: the fused up-sampling loader (ups_hi / ups_wi) exists for l4p_conv3d_k3 on the 16-bit engines, time axis not resized");
        return L4P_E_INVALID;
    }
    if (p.splitk > 1) {
        const int bk = 128 / es, nk = (p.K + bk - 1) / bk;
        if (p.epi != L4P_EPI_DENSE || !p.partial || p.splitk > nk || p.c_gr > 0) {
            l4p_set_error("gemm: split-K needs the dense epilogue, a partial buffer and splitk <= %d k-tiles", nk);
            return L4P_E_INVALID;
        }
    }
    return dtype == L4P_BF16 ? launch_gemm_bf16(mode, p, stream) : dtype == L4P_F16 ? launch_gemm_f16(mode, p, stream) : launch_gemm_f32(mode, p, stream);
}

int launch_gemm_group_bf16(const GemmParams* p, int n, hipStream_t stream);
int launch_gemm_group_f16(const GemmParams* p, int n, hipStream_t stream);
int launch_gemm_group(int dtype, const GemmParams* p, int n, hipStream_t stream) {
    if (!p || n < 1 || n > L4P_GEMM_GROUP_MAX) {
        l4p_set_error("gemm_group: 1 <= n <= %d descriptors", L4P_GEMM_GROUP_MAX);
        return L4P_E_INVALID;
    }
    if (is16(dtype)) {
        for (int i = 0; i < n; ++i) {  // the checks of launch_gemm that the fast path would skip
            if (p[i].M <= 0 || p[i].N <= 0 || p[i].K <= 0 || p[i].N % 8 || (p[i].K * 2) % 16 || (p[i].ldw * 2) % 16 || (p[i].lda * 2) % 16 ||
                p[i].splitk > 1)
                goto one_by_one;
        }
        return dtype == L4P_F16 ? launch_gemm_group_f16(p, n, stream) : launch_gemm_group_bf16(p, n, stream);
    }
one_by_one:
    for (int i = 0; i < n; ++i) {
        const int rc = launch_gemm(dtype, 0, p[i], stream);
        if (rc) return rc;
    }
    return 0;
}
