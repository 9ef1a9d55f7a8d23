"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two separate passes over bench.py)
into per-kernel mean bytes per dispatch.  gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE under-reports
16-byte-per-lane reads by 2x -> doubled here; WRITE_SIZE taken as reported.  Units of both counters: KB.
usage: python tools/pmc_traffic.py <fetch.csv> <write.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(conv_\w+_kernel<[^>]*>|k_\w+|conv_splitk_epilogue)", name)
    return m.group(1) if m else name[:60]


def agg(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fetch = 2.0 * 1024.0 * sf / nf if nf else 0.0
        write = 1024.0 * sw / nw if nw else 0.0
        out[k] = {"dispatches": nf or nw, "fetch_bytes_per_dispatch": fetch, "write_bytes_per_dispatch": write,
                  "hbm_bytes_per_dispatch": fetch + write}
    digest = None
    try:  # ties the profile to the kernel sources it was taken on (bench.py reports a stale profile as such)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        digest = bench.kernel_source_digest()
    except Exception:
        pass
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) over bench.py; FETCH_SIZE x2 (gfx950), KB -> bytes",
               "kernel_source_digest": digest, "kernels": out}, open(sys.argv[3], "w"), indent=1)
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_dispatch"] * kv[1]["dispatches"])[:8]
    for k, v in top:
        print("%-50s n=%5d  %.1f MB/dispatch" % (k, v["dispatches"], v["hbm_bytes_per_dispatch"] / 1e6))


if __name__ == "__main__":
    main()
