#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM-side traffic per launch.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and corrections as /opt/skills/guides (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores.  Infinity-Cache hits are counted (this is fabric-side traffic, an upper bound on HBM).
"""
import collections
import csv
import json
import re
import sys


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        d[name][0] += 1
        d[name][1] += float(r["Counter_Value"])
    return d


def main(fetch_csv, write_csv, out):
    f, w = agg(fetch_csv), agg(write_csv)
    res = {}
    for k in f:
        if not k.startswith(("conv_", "wgrad_", "bn_", "classifier", "crop", "momentum", "colsum", "stitch")):
            continue
        fb = f[k][1] / f[k][0] * 1024 * 2
        wb = (w[k][1] / w[k][0] * 1024) if k in w else 0.0
        res[k] = dict(launches=f[k][0], fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb),
                      traffic_bytes_per_launch=round(fb + wb))
    # the dominant kernel of bench.py's roofline: all launches of the forward / input-gradient conv kernel together
    # (exact fp32: the LDS-DMA form conv_dma_kernel and, for conv1's packed K-steps, conv_igemm_kernel; split-bf16: conv_split_dma_kernel)
    for name, fams in (("conv fwd+dgrad (all)", ("conv_dma_kernel", "conv_igemm_kernel")), ("conv_split_dma_kernel (all)", ("conv_split_dma",))):
        conv = [k for k in res if k.startswith(fams)]
        n = sum(res[k]["launches"] for k in conv)
        if n:
            res[name] = dict(launches=n, traffic_bytes_per_launch=round(sum(res[k]["traffic_bytes_per_launch"] * res[k]["launches"] for k in conv) / n))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k]["traffic_bytes_per_launch"]):
        print("%-48s %4d launches  %8.1f MB/launch" % (k, res[k]["launches"], res[k]["traffic_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
