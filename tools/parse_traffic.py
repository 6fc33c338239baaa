"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_workload.py into per-kernel HBM bytes.

Correction per MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
a wide coalesced streaming read -- calibrated here on the known-size copy probes in the same pass (16-byte and
8-byte lanes), and the read side of each kernel is scaled by the probe whose lane width matches its loads."""
import collections
import csv
import glob
import json
import sqlite3
import sys


def load(d, counter):
    """per-kernel values of one counter from a rocprofv3 --pmc run: CSV output if present, else the rocpd database"""
    acc = collections.defaultdict(list)
    csvs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if csvs:
        for r in csv.DictReader(open(csvs[0])):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return acc
    for fn in glob.glob(d + "/**/*_results.db", recursive=True):
        db = sqlite3.connect(fn)
        for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            acc[name].append(float(val))
    return acc


def main(fetch_dir, write_dir, out):
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    gib = float(1 << 30)

    def first(acc, key):
        for k, v in acc.items():
            if key in k:
                return sum(v) / len(v)
        return None

    cal = {}
    for lanes, key in ((16, "k_probe_copy16"), (8, "k_probe_copy8")):
        f, w = first(fe, key), first(wr, key)
        cal[lanes] = {"fetch_KiB": f, "write_KiB": w, "read_scale": gib / (f * 1024.0), "write_scale": gib / (w * 1024.0)}
    res = {"calibration": cal, "kernels": {}}
    # lane widths of the loads / stores (which copy probe scales the counter); kernels not listed use 8-byte lanes both ways
    lanes = {"k_spectrum": (8, 16), "k_mix_dec1": (16, 8), "k_mix_hb11_lean": (16, 8), "k_mix_hb11_bank": (16, 8)}

    def short(k):
        k = k.replace("void ", "").replace("pg::", "")
        for stop in "<(":
            if stop in k:
                k = k[:k.index(stop)]
        return k.strip()

    names = collections.defaultdict(list)
    for k in fe:
        if not short(k).startswith("k_probe"):
            names[short(k)].append(k)
    # (template instances of one kernel are pooled under its bare name, and every launch of the run counts: the workload scripts
    # run one route per kernel name; "k_spectrum" keeps its historical key for whichever 8192-bin instance ran)
    for name, full in sorted(names.items()):
        fv = [v for k in full for v in fe[k]]
        wv = [v for k in full for v in wr.get(k, [])]
        if not fv or not wv:
            continue
        lr, lw = next((v for key, v in lanes.items() if name.startswith(key)), (8, 8))
        f, w = sum(fv) / len(fv), sum(wv) / len(wv)
        rb = f * 1024.0 * cal[lr]["read_scale"]
        wb = w * 1024.0 * cal[lw]["write_scale"]
        res["kernels"][name] = {"launches": len(fv), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "read_bytes": rb, "write_bytes": wb, "hbm_bytes": rb + wb}
    for name in list(res["kernels"]):
        if name.startswith("k_spectrum_") and "k_spectrum" not in res["kernels"]:
            res["kernels"]["k_spectrum"] = dict(res["kernels"][name], instance=name)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
