"""Summarises the passes of tools/collect_pmc_valu.sh into one JSON file: per hot kernel the average counter values per
launch, the average duration from the kernel-trace pass of the same command, and the ratios that follow from the two --
vector instructions issued per SIMD cycle slot, cycles per vector instruction, LDS bank-conflict share.

python tools/pmc_valu_to_json.py <dir of collect_pmc_valu.sh> <out.json>"""
import collections
import csv
import glob
import json
import sys

SIMDS = 256 * 4
CLOCK_NOMINAL = 2.4e9
KERNELS = ("accumulate_kernel", "ntt_tile_kernel", "evalh_interp_kernel")


def short(name):
    return name.split("(")[0].split("<")[0].replace("void ", "").replace("zkhip::", "").strip()


def counters(directory):
    """{kernel: {counter: (average per launch, launches)}} over every counter_collection.csv under the directory"""
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[key] += float(r["Counter_Value"])
            cnt[key] += 1
    out = collections.defaultdict(dict)
    for (k, c), v in agg.items():
        out[k][c] = (v / cnt[(k, c)], cnt[(k, c)])
    return out


def durations(directory):
    out = {}
    for f in glob.glob(f"{directory}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return out


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    res = {"source": "rocprofv3 --pmc, one pass per counter group (tools/collect_pmc_valu.sh): `python3 bench.py --skip-cpu "
                     "--skip-extras --steps 12 --warmup 2` for accumulate_kernel and ntt_tile_kernel, `python3 "
                     "tools/evalh_bench.py` for evalh_interp_kernel; durations from a --kernel-trace --stats pass of the same "
                     "commands",
           "units": {"SQ_INSTS_*": "wave-level instructions per launch", "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*":
                     "as rocprofv3 reports them (summed over the chip; the wave-cycle family counts quad-cycles)",
                     "FETCH_SIZE / WRITE_SIZE": "KiB; FETCH_SIZE reads half the bytes of 16-byte-per-lane coalesced loads"},
           "kernels": {}}
    dur = {}
    for sub in ("bench_trace", "evalh_trace"):
        dur.update(durations(f"{d}/{sub}"))
    cnt = collections.defaultdict(dict)
    for sub in ("bench_SQ1", "bench_SQ2", "evalh_SQ1", "evalh_SQ2", "evalh_FETCH", "evalh_WRITE"):
        for k, v in counters(f"{d}/{sub}").items():
            cnt[k].update(v)
    for k in KERNELS:
        if k not in cnt or k not in dur:
            continue
        c = {name: v for name, (v, _) in cnt[k].items()}
        us = dur[k]["avg_us"]
        valu = c.get("SQ_INSTS_VALU", 0.0)
        slots = SIMDS * us * 1e-6 * CLOCK_NOMINAL / 4      # quarter-rate issue slots of the chip in one launch
        e = {"avg_us": us, "launches_traced": dur[k]["calls"], "counters_per_launch": c,
             # a wave64 vector instruction of the quarter-rate classes occupies its SIMD for four cycles: the chip has
             # 1024 SIMDs x 2.4 GHz / 4 = 614 G such slots a second
             "valu_issue_frac": valu / slots if slots else None,
             "cycles_per_valu_inst_at_2p4GHz": SIMDS * us * 1e-6 * CLOCK_NOMINAL / valu if valu else None}
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_WAVE_CYCLES"):
            e["wait_inst_lds_share_of_wave_cycles"] = c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in c:
            e["hbm_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
            e["hbm_bytes_note"] = "FETCH_SIZE x 2 (streaming 16-byte loads are tallied at half their size on gfx950) + WRITE_SIZE"
        res["kernels"][k] = e
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: {x: v[x] for x in v if x != "counters_per_launch"} for k, v in res["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
