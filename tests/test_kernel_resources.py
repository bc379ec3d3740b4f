"""Build-time guard (no GPU needed): the hot MFMA kernels must not spill.  A spill inside the 8-phase GEMM loop is not just
slow - the compiler waits `vmcnt(0)` for its scratch reload and thereby drains the LDS-DMA queue the loop is built around
(measured: -35 %)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _usage(src):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", f"-I{ROOT}/include",
           "-c", os.path.join(ROOT, "l4p_amd", "csrc", src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
        for key in ("VGPRs", "ScratchSize [bytes/lane]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                res[name][key] = int(m.group(1))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_hot_kernels_do_not_spill():
    gemm = _usage("gemm_bf16.hip")
    # (the shipped build: the measured-and-not-adopted kernels - gemm4w.hpp, the up-sampling loader of conv3_halo.hpp - are compiled
    #  only with PROBES=1 and must not be in this object)
    assert not any("gemm4w_kernel" in k or ("conv3_halo_kernel" in k and "ELb1EEv" in k) for k in gemm), sorted(gemm)
    hot = [k for k in gemm if "gemm8p_kernel" in k or ("gemm_kernel" in k and "ELb1E" in k) or
           ("conv3_halo_kernel" in k and k.endswith("ELb0EEv13l4p_gemm_desc"))]
    assert sum("conv3_halo_kernel" in k for k in hot) == 2, sorted(gemm)
    assert len(hot) >= 5, sorted(gemm)
    for k in hot:
        assert gemm[k]["ScratchSize [bytes/lane]"] == 0, (k, gemm[k])
        assert gemm[k]["VGPRs"] <= 256
    # the one-wave kernel of the tracker's token-side projections: an 18-step ring of fragments (216 registers) that must not spill, and
    # whose descriptor patches (row-grouped weights, the grouped launch's run-time descriptor choice) must stay in scalar registers
    sk = [k for k in gemm if "gemm_skinny" in k]
    assert len(sk) == 3, sorted(gemm)
    for k in sk:
        assert gemm[k]["ScratchSize [bytes/lane]"] == 0, (k, gemm[k])
        assert gemm[k]["VGPRs"] <= 256
    attn = _usage("attention.hip")
    hot = [k for k in attn if "DF16b" in k]
    assert hot
    for k in hot:
        assert attn[k]["ScratchSize [bytes/lane]"] == 0, (k, attn[k])
    # the one-wave-per-SIMD attention: its state (S ping-pong, Q, P, O in the accumulator file) must stay in registers - the kernel
    # whose closures once exceeded the optimiser's scalar-replacement limit and went to scratch wholesale (1344 bytes per lane)
    a64 = _usage("attention64.hip")
    assert len(a64) == 4, sorted(a64)
    for k in a64:
        assert a64[k]["ScratchSize [bytes/lane]"] == 0, (k, a64[k])
        assert a64[k]["VGPRs"] <= 256
