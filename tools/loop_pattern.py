#!/usr/bin/env python3
"""Instruction pattern of the steady-state stage loop of a wave kernel (first backward branch behind the first MFMA):
   tools/loop_pattern.py file.o 'mangled-name regex'
M = MFMA, D = LDS-DMA gather, r = ds_read, v = VALU, g = other global load, S = scratch, w = s_waitcnt, . = scalar / nop"""
import collections
import re
import subprocess
import sys
import tempfile

L = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    t = tempfile.mkdtemp()
    subprocess.run([f"{L}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"{t}/x.fatbin"], check=True)
    subprocess.run([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/x.fatbin",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={t}/x.co"], check=True, capture_output=True)
    return subprocess.run([f"{L}/llvm-objdump", "-d", f"{t}/x.co"], check=True, capture_output=True, text=True).stdout


def kind(ln):
    return ("M" if "mfma" in ln else "D" if "global_load_lds" in ln else "r" if ln.startswith("ds_read") else "v" if ln.startswith("v_")
            else "g" if ln.startswith("global_load") else "S" if ln.startswith("scratch_") else "w" if ln.startswith("s_waitcnt") else ".")


def main():
    txt = disassemble(sys.argv[1])
    for m in re.finditer(r"\n[0-9a-f]+ <(\S+)>:\n(.*?)(?=\n[0-9a-f]+ <|\Z)", txt, re.S):
        if not re.search(sys.argv[2], m.group(1)):
            continue
        lines = [ln.strip().split("//")[0].strip() for ln in m.group(2).splitlines()]
        first = end = None
        for i, ln in enumerate(lines):
            if "mfma" in ln and first is None:
                first = i
            b = re.match(r"s_cbranch_\w+ (\d+)", ln)
            if b and int(b.group(1)) > 60000 and first is not None:
                end = i
                break
        if end is None:
            continue
        start = max(j for j in range(first) if lines[j].startswith("s_waitcnt vmcnt(0)"))
        body = lines[start:end + 1]
        seq = "".join(kind(ln) for ln in body)
        print(m.group(1))
        print(" ", dict(collections.Counter(seq)), "instructions", len(body))
        print(" ", seq)
        if len(sys.argv) > 3:
            print("\n".join(body))


if __name__ == "__main__":
    main()
