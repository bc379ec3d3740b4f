"""Instruction mix per barrier-delimited segment of one kernel of a .hip file (tuning aid).
usage: python tools/isa_segments.py l4p_amd/csrc/attention.hip <mangled-name-substring> [extra hipcc flags...]"""
import collections
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
asm = "/tmp/isa_segments.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                "-I/root/repo/include", "-S", "--cuda-device-only", src, "-o", asm] + sys.argv[3:], check=True,
               stderr=subprocess.DEVNULL)
s = open(asm).read()
m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):", s, re.M)
i = m.start()
j = s.index(".end_amdhsa_kernel", i)
print(m.group(1))
for key in ("vgpr_count", "next_free_vgpr", "accum_offset", "scratch", "private_segment_fixed_size"):
    mm = re.search(r"\." + key + r"[ :]+(\d+)", s[i:j + 3000])
    if mm:
        print(f"  .{key} {mm.group(1)}")
seg, cur = [], []
for l in s[i:j].split("\n"):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    cur.append(l)
    if l.startswith("s_barrier"):
        seg.append(cur)
        cur = []
seg.append(cur)
for k, c in enumerate(seg):
    cnt = collections.Counter()
    for l in c:
        op = l.split()[0]
        if op.endswith(":"):
            continue
        if op.startswith("v_mfma"):
            cnt["mfma"] += 1
        elif op.startswith("v_exp"):
            cnt["exp"] += 1
        elif op.startswith("ds_"):
            cnt["ds"] += 1
        elif op.startswith("global_load") or op.startswith("buffer_load"):
            cnt["gload"] += 1
        elif op.startswith("scratch_"):
            cnt["scratch"] += 1
        elif op.startswith("v_"):
            cnt["valu:" + op] += 1
        elif op.startswith("s_waitcnt"):
            cnt["wait"] += 1
        elif op.startswith("s_nop"):
            cnt["nop"] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
    valu = sum(v for k2, v in cnt.items() if k2.startswith("valu:"))
    print(k, len(c), "mfma", cnt["mfma"], "exp", cnt["exp"], "valu", valu, "ds", cnt["ds"], "gl", cnt["gload"], "wait", cnt["wait"],
          "nop", cnt["nop"], "salu", cnt["salu"], "scratch", cnt["scratch"])
    if cnt["mfma"] >= 12:
        print("    ", dict(sorted(((k2[5:], v) for k2, v in cnt.items() if k2.startswith("valu:")), key=lambda kv: -kv[1])))
