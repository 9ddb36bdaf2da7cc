#!/usr/bin/env python3
"""Print the instruction-class pattern of every MFMA-heavy basic block of one kernel:
   tools/loop_schedule.py prnn _ZN5rnnwf16prnn_flip_kernelIfLi3ELi4EEEvNS_8PrnnArgsE
M mfma, v valu, t transcendental, d LDS, g global, w s_waitcnt, n s_nop, s other scalar."""
import re, subprocess, sys
tu, name = sys.argv[1], sys.argv[2]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value", "-ffp-contract=fast",
                "-mllvm", "-amdgpu-mfma-vgpr-form", "--offload-device-only", "-S",
                "rnnwavefunctions_amd/csrc/%s.hip" % tu, "-o", "/tmp/%s.s" % tu], check=True, capture_output=True)
s = open("/tmp/%s.s" % tu).read()
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
blocks, cur, label = [], [], "entry"
for line in s[i:j].split("\n"):
    if re.match(r"^\.LBB\d+_\d+:", line):
        blocks.append((label, cur)); cur = []; label = line.strip()
    else:
        cur.append(line.strip())
blocks.append((label, cur))
TR = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f64", "v_rsq_f64")
for lab, cur in blocks:
    ins = [l for l in cur if l and not l.startswith(";") and not l.startswith(".")]
    nm = sum("v_mfma" in l for l in ins)
    if nm < 50:
        continue
    p = ""
    for l in ins:
        op = l.split()[0]
        p += ("M" if "mfma" in op else "t" if op.startswith(TR) else "v" if op.startswith("v_") else "d" if op.startswith("ds_")
              else "w" if op.startswith("s_waitcnt") else "n" if op.startswith("s_nop") else "s" if op.startswith("s_")
              else "g" if op.startswith(("global", "buffer", "flat")) else "?")
    print(lab[:70], "instr=%d mfma=%d valu=%d trans=%d ds=%d waitcnt=%d" % (len(ins), nm, p.count("v"), p.count("t"), p.count("d"), p.count("w")))
    for k in range(0, len(p), 130):
        print("   ", p[k:k + 130])
m = re.search(re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", s[j:], re.S)
vg = re.search(r"\.name:\s+" + re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", s, re.S)
print("vgpr_count", vg.group(1) if vg else "?")
