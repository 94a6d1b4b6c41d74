#!/bin/bash
# HBM traffic per launch of the product kernels from the PMC counters (MI355X_MICROARCH.md "HBM [CDNA4]"):
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains), units KB; FETCH_SIZE calibrated on known
# byte counts in this path's own access widths (16 B/lane: guide says x2; 12 B/lane: measured here).
# usage: tools/traffic.sh [outdir]   -> <outdir>/traffic.json
set -u
OUT=${1:-gpurun_out/traffic}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-events ${AGX_TRAFFIC_BENCH_ARGS:-}"
BUILD=$(python3 -c "import sys; sys.path.insert(0, 'active-gym_amd'); from active_gym import _native as n; print(n.build_info())")
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -- tools/membench cal > $OUT/cal_$c.log 2>&1 || echo "cal $c failed"
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/bench_$c -- $BENCH > $OUT/bench_$c.log 2>&1 || echo "bench $c failed"
done
python3 - "$OUT" "$BUILD" "${AGX_TRAFFIC_BENCH_ARGS:-}" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def collect(prefix):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{prefix}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) * 1024.0 for c, v in d.items()} for k, d in agg.items()}     # KB -> bytes
N = 1024
cal = collect("cal")
known = {"k_lin": N * 2 * 210 * 480, "k_lin3": (N * 2 * 210 * 480 // 12) * 12, "k_x3": N * 2 * 168 * 480, "k_st": N * 4 * 1764 * 16}
res = {"unit": "bytes per launch", "build": sys.argv[2], "bench_args": sys.argv[3], "calibration": {}, "kernels": {}}
f16 = f12 = None
for k, d in cal.items():
    short = k.split("::")[-1].strip()
    for name in known:
        if short == name or short.startswith(name + "<"):
            e = {"known_bytes": known[name], **{c: v for c, v in d.items()}}
            if "FETCH_SIZE" in d and name != "k_st":
                e["fetch_factor"] = known[name] / d["FETCH_SIZE"]
            if "WRITE_SIZE" in d and name == "k_st":
                e["write_factor"] = known[name] / d["WRITE_SIZE"]
            res["calibration"][name] = e
f16 = res["calibration"].get("k_lin", {}).get("fetch_factor")
f12 = res["calibration"].get("k_lin3", {}).get("fetch_factor")
wf = res["calibration"].get("k_st", {}).get("write_factor")
res["factors"] = {"fetch_16B_lane": f16, "fetch_12B_lane": f12, "write_16B_lane": wf}
for k, d in collect("bench").items():
    if "agx::" not in k:
        continue
    short = k.split("agx::")[1].split("<")[0].split("(")[0]
    if short.startswith("k_ingest"):                    # k_ingest<256>, k_ingest_full12: one family, one key
        short = "k_ingest"
    ff = f12 if short == "k_ingest" else f16            # K1 loads 12 B per lane; K2-K4 read the ring in dwords/x4
    e = {"kernel_name": k, "FETCH_SIZE_raw": d.get("FETCH_SIZE"), "WRITE_SIZE_raw": d.get("WRITE_SIZE"), "fetch_factor_used": ff}
    if ff and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        e["fetch_bytes"] = d["FETCH_SIZE"] * ff
        e["write_bytes"] = d["WRITE_SIZE"] * (wf or 1.0)
        e["traffic"] = e["fetch_bytes"] + e["write_bytes"]
    if short in ("k_fovea_peripheral3", "k_fovea_peripheral2"):
        short = "k_fovea_peripheral"
    if short in ("k_fovea_flexible3", "k_fovea_flexible2"):
        short = "k_fovea_flexible"
    res["kernels"][short] = e
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
