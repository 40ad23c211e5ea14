#!/usr/bin/env python3
"""HBM-side traffic per kernel launch from rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py).

Collect (separate passes, counters only -- FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE -d <out>/fetch -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --profile-steps 0
  rocprofv3 --pmc WRITE_SIZE -d <out>/write -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --profile-steps 0
Aggregate:
  python tools/pmc_traffic.py <out>/fetch <out>/write profiles/pmc_traffic.json

Units / corrections (guide, HBM section): both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-B requests
of wide streaming reads at 64 B, so bytes read = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-B-per-lane stores and
float atomics.  bench.py computes traffic = (2 * fetch + write) KB per launch.  Infinity-Cache hits are counted
(the counters sit on the L2's fabric side), so this is L2-miss traffic, an upper bound of HBM traffic.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under %s" % d)
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            name = name.replace("(anonymous namespace)::", "").split("(")[0].strip()
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        kernels[k] = {"launches": max(nf, nw), "fetch_KB_per_launch": sf / nf if nf else 0.0,
                      "write_KB_per_launch": sw / nw if nw else 0.0}
    note = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB from separate rocprofv3 --pmc passes of `bench.py --steps 4 "
            "--warmup 2` (tools/pmc_traffic.py); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide streaming "
            "reads at half); Infinity-Cache hits are included (fabric-side counters)")
    # where and when the passes ran (bench.py copies this into roofline.traffic_source, so a bench line says whether its
    # traffic figure comes from the box it ran on)
    import socket
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vpd_amd.boxid import gpu_unique_id
    # (every container of the pool is called the same: the GPU's unique_id from the KFD topology names the box)
    source = {"tag": os.environ.get("VPD_PROFILE_TAG", ""), "host": socket.gethostname(), "gpu_unique_id": gpu_unique_id(0),
              "collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
              "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 4 --warmup 2 --repeats 1 --profile-steps 0 --no-cpu-baseline --no-apply"}
    json.dump({"note": note, "source": source, "kernels": kernels}, open(out, "w"), indent=1)
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["launches"] * (2 * kv[1]["fetch_KB_per_launch"] + kv[1]["write_KB_per_launch"]))
    for k, v in top[:12]:
        print("%-70s %5d  rd %9.1f MB  wr %8.1f MB" % (k[:70], v["launches"], 2 * v["fetch_KB_per_launch"] / 1024, v["write_KB_per_launch"] / 1024))


if __name__ == "__main__":
    main()
