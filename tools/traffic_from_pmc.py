#!/usr/bin/env python3
"""Turns a PMC summary (tools/pmc_summary.py) into an entry of profiles/traffic.json: HBM bytes per launch of
every rendering kernel (the non-counting instantiation, `<false>` or `<false, placement>`) for ONE configuration.

  traffic_from_pmc.py <summary.json> <config> [profiles/traffic.json] [bench.json of the same command]

The bench line of the profiled command tells how many pixel-samples one batch held ("batch": {"pixel_samples"}); it is
stored as _batch_pixel_samples, so that bytes per launch can be related to pixel-samples.

The file maps configuration -> kernel -> bytes per launch; other configurations' entries are kept.

  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE / WRITE_SIZE are reported in KiB. On gfx950 FETCH_SIZE tallies a 128-B fabric read as
64 B, so a wide coalesced stream reads twice the reported bytes; the factor 2 is applied to every
kernel here, which makes the figure an upper bound for the kernels whose requests are mostly <= 64 B
(scattered vertex gathers, tree nodes). The raw counters are kept beside the corrected value.
"""
import json
import sys


def main():
    src, config = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
    bench = sys.argv[4] if len(sys.argv) > 4 else None
    d = json.load(open(src))
    out, detail = {}, {}
    for name, c in d.items():
        # the non-counting instantiations: k<false> or k<false, placement>; and the two rendering kernels without a
        # counting form (their launches in the short counting pass of the bench are in the average)
        if ("<false" not in name and name not in ("k_pair_compact", "k_commit_batch")) or "FETCH_SIZE" not in c:
            continue
        n = c["dispatches"]
        fetch_kib, write_kib = c["FETCH_SIZE"] / n, c.get("WRITE_SIZE", 0.0) / n
        k = name.split("<")[0]
        out[k] = (2.0 * fetch_kib + write_kib) * 1024.0
        detail[k] = {"dispatches": n, "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                     "uncorrected_bytes_per_launch": (fetch_kib + write_kib) * 1024.0}
    out["_detail"] = detail
    if bench:
        line = [l for l in open(bench).read().strip().splitlines() if l.startswith("{")][-1]
        out["_batch_pixel_samples"] = json.loads(line)["batch"]["pixel_samples"]
    out["_formula"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; FETCH doubled for gfx950 (upper bound for <=64-B requests)"
    try:
        allc = json.load(open(dst))
        if not all(isinstance(v, dict) for v in allc.values()):
            allc = {}
    except Exception:
        allc = {}
    allc[config] = out
    json.dump(allc, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        if not k.startswith("_"):
            print("%-20s %10.1f MB/launch" % (k, v / 1e6))


if __name__ == "__main__":
    main()
