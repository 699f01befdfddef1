#!/usr/bin/env python3
"""profiles/<round>/<tag>/pmc_fetch.txt + pmc_write.txt -> one traffic entry (JSON on stdout): HBM-side bytes per
launch of the dominant half-iteration kernel of that run, the X-side and Theta-side launches separately.
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (it
tallies 128-byte requests at 64 bytes).  `kernel` is the name as the library reports it (cumf_last_kernel_name:
rocprofv3's name without the leading "void " and the parameter list) -- bench.py replays an entry only for the
kernel it dispatched.  tools/merge_traffic.py collects the entries of a round into profiles/traffic.json."""
import json
import re
import sys

GRAM_KERNELS = ("als_wave_kernel", "als_wave_multi_kernel", "als_item_kernel")


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


def norm(name):
    return name[5:] if name.startswith("void ") else name


def main():
    d = sys.argv[1]
    fetch, write = parse(f"{d}/pmc_fetch.txt"), parse(f"{d}/pmc_write.txt")
    cand = [k for k in fetch if any(g in k[0] for g in GRAM_KERNELS)]
    if not cand:
        raise SystemExit(f"{d}: no Gram kernel in pmc_fetch.txt")
    # the dominant kernel FAMILY = the Gram kernel with the most fetched bytes over its launches; the two instances of
    # the wave kernel that differ only in the last template argument (WHOLE: the plan has no chunked rows -- the
    # Theta side of the Netflix shape -- or it has -- the X side) are one family
    # -- and, round 6, in the arithmetic argument in front of it (0: in-kernel split, the X side, whose gather table is
    # HBM-resident; 3 / 2: the pre-split table of a cache-resident side, the Theta side) -- are one family
    def family(n):
        n = re.sub(r"(als_wave_kernel<\d+, \d+, \d+), \d+, (true|false)>$", r"\1>", n)
        return re.sub(r",\s*(true|false)>$", ">", n)

    by_name = {}
    for k in cand:
        by_name[family(k[0])] = by_name.get(family(k[0]), 0.0) + fetch[k]["FETCH_SIZE"]
    fam = max(by_name, key=by_name.get)
    main_keys = sorted((k for k in cand if family(k[0]) == fam), key=lambda k: k[1])  # X side: fewer items than Theta has rows
    main_keys = [main_keys[0], main_keys[-1]] if len(main_keys) > 1 else main_keys
    name = main_keys[0][0]   # the X-side instance: what bench.py reports as roofline.kernel
    tag = d.rstrip("/").split("gpurun_out/")[-1].replace("profiles_", "profiles/")
    res = {"kernel": norm(name),
           "source": f"{tag}/pmc_fetch.txt + pmc_write.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per "
                     "dispatch; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per the gfx950 note of "
                     "MI355X_MICROARCH.md)"}
    sides = {}
    for side, k in zip(("x_side", "theta_side"), main_keys[:2]):
        f_kib, w_kib = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", 0.0)
        sides[side] = {"kernel": norm(k[0]), "grid": k[1], "fetch_size_kib": f_kib, "write_size_kib": w_kib,
                       "bytes_per_launch": (2 * f_kib + w_kib) * 1024.0}
    res.update(sides)
    if len(sides) == 2:
        res["bytes_per_launch"] = 0.5 * (sides["x_side"]["bytes_per_launch"] + sides["theta_side"]["bytes_per_launch"])
    elif sides:
        # one launch only (the slab mode's partial-Gram kernel: the X side dispatches another kernel family)
        res["launch"] = sides.pop("x_side")
        res.pop("x_side", None)
        res["bytes_per_launch"] = res["launch"]["bytes_per_launch"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
