#!/bin/bash
# Registers / spills / scratch and the instruction mix of every kernel in a gfx950 host object (or .so):
#   tools/kernel_regs.sh file.o ['name regex'] [--mix]
O=$1; F=${2:-.}; MIX=${3:-}
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin $O $T/x.fatbin
$L/clang-offload-bundler --unbundle --type=o --input=$T/x.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/x.co 2>/dev/null
$L/llvm-readelf --notes $T/x.co > $T/notes.txt
[ -n "$MIX" ] && $L/llvm-objdump -d $T/x.co > $T/dis.txt
python3 - "$T" "$F" "$MIX" <<'PY'
import re, subprocess, sys, collections
T, F, MIX = sys.argv[1], sys.argv[2], sys.argv[3]
txt = open(T + "/notes.txt").read()
def g(k, b):
    m = re.search(r"\." + k + r":\s+(\S+)", b)
    return m.group(1) if m else "?"
dis = open(T + "/dis.txt").read() if MIX else ""
for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    blk = ".agpr_count" + blk
    n = g("name", blk)
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    if not re.search(F, d):
        continue
    print(d[:100].ljust(100), "vgpr", g("vgpr_count", blk), "agpr", g("agpr_count", blk), "sgpr", g("sgpr_count", blk), "spill",
          g("vgpr_spill_count", blk), "scratch", g("private_segment_fixed_size", blk))
    if MIX:
        m = re.search(r"\n[0-9a-f]+ <" + re.escape(n) + r">:\n(.*?)(?=\n[0-9a-f]+ <|\Z)", dis, re.S)
        if m:
            c = collections.Counter()
            for ln in m.group(1).splitlines():
                p = ln.strip().split()
                if not p: continue
                op = p[0]
                k = ("mfma_bf16" if "mfma" in op and "bf16" in op else "mfma_f32" if "mfma" in op else "dma" if "global_load_lds" in op else
                     "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "ds_bperm" if "bpermute" in op else "ds" if op.startswith("ds_") else
                     "s_nop" if op == "s_nop" else "s_wait" if op.startswith("s_waitcnt") else "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "other")
                c[k] += 1
            print("    mix:", dict(sorted(c.items())), "total", sum(c.values()))
PY
rm -rf $T
