"""Condenses gpurun_out/r02prof (profiles/r02_profile.sh) into the small files kept under profiles/:
   r02_kernel_stats_<cfg>.csv   top rows of rocprofv3 --kernel-trace --stats
   r02_traffic.json             HBM bytes per stepper launch from the FETCH_SIZE / WRITE_SIZE passes
(the figures are copied into profiles/traffic.json by hand once checked)."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
for cfg in ("config3", "config4", "config2"):
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum"):
        files = glob.glob(os.path.join(out, f"pmc_{cfg}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        vals = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                if "crb_step_lean_kernel" not in row["Kernel_Name"]:
                    continue
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                per["kernel"] = row["Kernel_Name"]
        for c, v in vals.items():
            # all stepper dispatches of a run are identical launches (spin-up, warmup and timed ones): mean and spread
            per[c] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        # rocprofv3 reports both in KiB; gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x2
        per["hbm_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"]["mean"] + per["WRITE_SIZE"]["mean"]) * 1024.0
    res[cfg] = per
json.dump(res, open(os.path.join(out, "r02_traffic.json"), "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in res.items()}))
for f in glob.glob(os.path.join(out, "kernel_stats_*.csv")):
    rows = open(f).read().splitlines()
    open(os.path.join(out, "r02_" + os.path.basename(f)), "w").write("\n".join(rows[:9]) + "\n")
