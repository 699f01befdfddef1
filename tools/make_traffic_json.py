#!/usr/bin/env python3
"""profiles/<round>/<tag>/pmc_fetch.txt + pmc_write.txt -> traffic.json: HBM-side bytes per launch of
the dominant half-iteration kernel, the X-side and Theta-side launches separately.
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
gfx950 (it tallies 128-byte requests at 64 bytes)."""
import json
import re
import sys


def parse(path):
    out, key = {}, None
    for line in open(path):
        m = re.match(r"(\S.*?)\s+grid=(\d+)", line)
        if m:
            key = (m.group(1).strip(), int(m.group(2)))
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=(\S+)", line)
        if m and key:
            out.setdefault(key, {})[m.group(1)] = float(m.group(3))
    return out


d = sys.argv[1]
fetch, write = parse(f"{d}/pmc_fetch.txt"), parse(f"{d}/pmc_write.txt")
main = [k for k in fetch if "als_wave_kernel" in k[0] or "als_item_kernel" in k[0]]
main.sort(key=lambda k: k[1])  # the X side has fewer items (rows are chunked) than the Theta side has rows
res = {"source": f"{d.split('gpurun_out/')[-1]}/pmc_fetch.txt + pmc_write.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                 "separate passes, per dispatch; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per the "
                 "gfx950 note of MI355X_MICROARCH.md)"}
sides = {}
for name, k in zip(("x_side", "theta_side"), main[:2]):
    f_kib, w_kib = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", 0.0)
    sides[name] = {"kernel": k[0], "grid": k[1], "fetch_size_kib": f_kib, "write_size_kib": w_kib,
                   "bytes_per_launch": (2 * f_kib + w_kib) * 1024.0}
res.update(sides)
if len(sides) == 2:
    res["bytes_per_launch"] = 0.5 * (sides["x_side"]["bytes_per_launch"] + sides["theta_side"]["bytes_per_launch"])
print(json.dumps(res, indent=1))
