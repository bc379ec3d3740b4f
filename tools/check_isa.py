#!/usr/bin/env python3
"""ISA lint of libl4p_hip.so: no instruction form that a gfx950 hardware interaction is known to corrupt.

The interaction (found in round 5; reproducer tools/probes/mfma_valu_probe.py, write-up in DESIGN.md): while ANOTHER wave of the same
SIMD issues MFMAs (16x16x32 f16 / bf16, 32x32x16 bf16 measured), a packed-FP32 VALU instruction whose LOW result lane takes the HIGH
dword of src1 -

    v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32  ...  op_sel:[x,1(,y)]      with src1 != src0

- computes its low result dword with src1 read as 0 in lanes 48..63 (the instruction's fourth pass).  Every other form measured
(no op_sel, op_sel on src0 or src2, op_sel_hi anything, v_pk_mov_b32, packed f16, f64, DPP, plain VALU) is unaffected.  The compiler
picks the form on its own when it folds a swizzle into a packed multiply; kernels that got it are compiled without packed FP32
(L4P_NO_PK_F32 in csrc/common.hpp).  This tool disassembles every gfx950 code object of the library and lists offenders; the CPU
test suite runs it (tests/test_host_cpu.py), so a source change that re-introduces the form fails the build check.

Scope: the library's own code objects.  The only third-party kernels the product path runs beside the library's MFMA kernels (side
streams, L4P_HEAD_STREAMS / L4P_TRACK_STREAMS) are torch's copy / fill / index kernels - integer and byte moves and float fills, no
packed-FP32 arithmetic (tests/test_stream_overlap_gpu.py holds the schedules bit for bit); `python tools/check_isa.py <any .so / .co>`
lints another code object the same way.

  python tools/check_isa.py [path/to/libl4p_hip.so]      exit status 1 if an offending instruction exists
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile



def _find_llvm_bin() -> str:
    """Directory holding llvm-objdump: $ROCM_PATH / $HIP_PATH / `hipconfig --rocmpath` / /opt/rocm, then whatever is on PATH."""
    cands = [os.environ.get(k) for k in ("ROCM_PATH", "HIP_PATH")]
    try:
        cands.append(subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True, timeout=20).stdout.strip())
    except (OSError, subprocess.SubprocessError):
        pass
    cands.append("/opt/rocm")
    for c in cands:
        if c and os.path.exists(os.path.join(c, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(c, "lib", "llvm", "bin")
    w = shutil.which("llvm-objdump")
    return os.path.dirname(w) if w else "/opt/rocm/lib/llvm/bin"


LLVM = _find_llvm_bin()
PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\s+(.*)$")
OPSEL = re.compile(r"\bop_sel:\[([01](?:,[01])+)\]")


def offending(line: str):
    """-> description if the instruction is of the affected form, else None."""
    m = PK.search(line)
    if not m:
        return None
    rest = m.group(2)
    s = OPSEL.search(rest)
    if not s:
        return None
    sel = s.group(1).split(",")
    if len(sel) < 2 or sel[1] != "1":
        return None
    ops = [o.strip() for o in rest.split(" op_sel")[0].split(",")]
    # the bracketed register ranges contain commas-free "v[a:b]" tokens; operands: dst, src0, src1(, src2)
    if len(ops) >= 3 and ops[1] == ops[2]:
        return None  # src0 == src1: measured unaffected (one operand read serves both)
    return f"{m.group(1)} {rest.strip()}"


def scan(lib: str):
    """-> list of (kernel symbol, instruction text)."""
    tmp = tempfile.mkdtemp(prefix="l4p_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not cos:
            raise RuntimeError(f"no gfx950 code object found in {lib}")
        hits, ninstr = [], 0
        for co in cos:
            out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
            sym = "?"
            for line in out.splitlines():
                if line.endswith(">:"):
                    sym = line.split("<")[-1][:-2]
                    continue
                if "v_pk_" in line:
                    ninstr += 1
                    text = line.split("//")[0].strip()
                    d = offending(text)
                    if d:
                        hits.append((sym, d))
        return hits, ninstr, len(cos)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main() -> int:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "l4p_amd", "lib", "libl4p_hip.so")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        # (degrade to a warning: a host without the ROCm LLVM tools can still link; the CPU test suite skips its lint test likewise)
        print(f"check_isa: llvm-objdump not found (ROCM_PATH / hipconfig / PATH) - ISA lint of {lib} SKIPPED", file=sys.stderr)
        return 0
    hits, ninstr, nco = scan(lib)
    print(f"{lib}: {nco} code objects, {ninstr} packed instructions scanned, {len(hits)} of the affected form")
    for sym, d in hits:
        try:
            sym = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip() or sym
        except OSError:
            pass
        print(f"  {sym[:140]}\n      {d}")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
