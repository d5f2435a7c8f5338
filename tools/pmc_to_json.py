"""Summarises the passes of tools/collect_pmc.sh into one JSON file: per kernel the average FETCH_SIZE / WRITE_SIZE per
launch, the calibration factors that tools/gather_probe gives for each access shape (counter bytes / bytes really moved),
and the corrected HBM bytes per launch of the two kernels the benchmark's roofline objects are about.

python tools/pmc_to_json.py <dir of collect_pmc.sh> <out.json>"""
import collections
import csv
import glob
import json
import sys


def per_kernel(directory, counter):
    """{kernel name (up to the first '<' or '('): (average counter value per launch in KiB, launches)}"""
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("zkhip::", "").strip()
            agg[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return {k: (agg[k] / cnt[k], cnt[k]) for k in agg}


def durations(directory):
    out = {}
    for f in glob.glob(f"{directory}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Name"].split("(")[0].split("<")[0].replace("void ", "").replace("zkhip::", "").strip()
            out[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return out


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(f"{d}/bench_FETCH_SIZE", "FETCH_SIZE"), per_kernel(f"{d}/bench_WRITE_SIZE", "WRITE_SIZE")
    pf, pw = per_kernel(f"{d}/probe_FETCH_SIZE", "FETCH_SIZE"), per_kernel(f"{d}/probe_WRITE_SIZE", "WRITE_SIZE")
    known = json.loads(open(f"{d}/probe_FETCH_SIZE.log").read().strip().splitlines()[-1])
    KIB = 1024.0
    calib = {}
    for shape in ("stream32", "stream4", "gather64"):
        if shape in pf:
            calib[shape] = {"counter_bytes": pf[shape][0] * KIB, "true_bytes": known[f"{shape}_bytes"],
                            "counter_over_true": pf[shape][0] * KIB / known[f"{shape}_bytes"]}
    if "store32" in pw:
        calib["store32"] = {"counter_bytes": pw["store32"][0] * KIB, "true_bytes": known["store32_bytes"],
                            "counter_over_true": pw["store32"][0] * KIB / known["store32_bytes"]}
    dur = durations(f"{d}/bench_trace")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, over `python3 bench.py --skip-cpu --skip-extras "
                     "--steps 12 --warmup 2`; calibration from tools/gather_probe (same passes)",
           "calibration": calib, "kernels_KiB_per_launch": {}, "kernel_avg_us": dur}
    for k in sorted(set(fetch) | set(write)):
        res["kernels_KiB_per_launch"][k] = {"FETCH_SIZE": fetch.get(k, (None, 0))[0], "WRITE_SIZE": write.get(k, (None, 0))[0],
                                            "launches": fetch.get(k, write.get(k))[1]}
    g = calib.get("gather64", {}).get("counter_over_true", 1.0)
    s32 = calib.get("stream32", {}).get("counter_over_true", 0.5)
    if "accumulate_kernel" in fetch:
        f_b, w_b = fetch["accumulate_kernel"][0] * KIB, write.get("accumulate_kernel", (0, 0))[0] * KIB
        res["accumulate_kernel"] = {
            "fetch_counter_bytes": f_b, "write_counter_bytes": w_b,
            "fetch_correction": 1.0 / g, "fetch_correction_from": "gather64 (64-byte gathers of table points dominate its reads)",
            "bytes_per_launch": f_b / g + w_b}
    if "ntt_tile_kernel" in fetch:
        f_b, w_b = fetch["ntt_tile_kernel"][0] * KIB, write.get("ntt_tile_kernel", (0, 0))[0] * KIB
        passes = 3
        res["ntt_tile_kernel"] = {
            "fetch_counter_bytes_per_launch": f_b, "write_counter_bytes_per_launch": w_b, "launches_per_transform": passes,
            "fetch_correction": 1.0 / s32, "fetch_correction_from": "stream32 (32 contiguous bytes per lane)",
            "bytes_per_transform": passes * (f_b / s32 + w_b)}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("calibration", "accumulate_kernel", "ntt_tile_kernel") if k in res}, indent=1))


if __name__ == "__main__":
    main()
