"""Condenses gpurun_out/r03prof (profiles/r03_profile.sh) into the small files kept under profiles/:
   r03_kernel_stats_<cfg>.csv   top rows of rocprofv3 --kernel-trace --stats
   r03_bench_<cfg>.json         the bench line of the traced run
   r03_traffic_pmc.json         HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, per kernel
and prints the entries for profiles/traffic.json (stamped with the hash of the kernel sources they were measured on)."""
import csv, glob, hashlib, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
KERNEL = {"config3": "crb_step_lean_kernel", "config4": "crb_step_lean_kernel", "config2": "crb_step_lean_kernel", "config5": "crb_loop_kernel"}
LEAN_SRC = ["continuum-robot_amd/csrc/crb_lean.h", "continuum-robot_amd/csrc/crb_lean.hip", "continuum-robot_amd/csrc/crb_math.h",
            "continuum-robot_amd/csrc/crb_generic.h"]
LOOP_SRC = ["continuum-robot_amd/csrc/crb_loop.h", "continuum-robot_amd/csrc/crb_loop.hip", "continuum-robot_amd/csrc/crb_lean.h",
            "continuum-robot_amd/csrc/crb_math.h", "continuum-robot_amd/csrc/crb_generic.h"]


def source_hash(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


res = {}
for cfg, kern in KERNEL.items():
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = {}
        for f in glob.glob(os.path.join(out, f"pmc_{cfg}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern not in row["Kernel_Name"]:
                    continue
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                per["kernel"] = row["Kernel_Name"]
        for c, v in vals.items():
            per[c] = {"mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v), "dispatches": len(v)}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        # rocprofv3 reports both in KiB; gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x2
        per["hbm_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"]["mean_KiB"] + per["WRITE_SIZE"]["mean_KiB"]) * 1024.0
        per["hbm_bytes_per_launch_uncorrected"] = (per["FETCH_SIZE"]["mean_KiB"] + per["WRITE_SIZE"]["mean_KiB"]) * 1024.0
    src = LOOP_SRC if cfg == "config5" else LEAN_SRC
    per["sources"] = src
    per["sources_sha256_16"] = source_hash(src)
    res[cfg] = per
json.dump(res, open(os.path.join(ROOT, "profiles", "r03_traffic_pmc.json"), "w"), indent=1)
print(json.dumps({k: (v.get("hbm_bytes_per_launch"), v.get("sources_sha256_16")) for k, v in res.items()}))
for f in glob.glob(os.path.join(out, "kernel_stats_*.csv")):
    rows = open(f).read().splitlines()
    open(os.path.join(ROOT, "profiles", "r03_" + os.path.basename(f)), "w").write("\n".join(rows[:9]) + "\n")
for f in glob.glob(os.path.join(out, "bench_*.json")):
    line = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if line:
        name = os.path.basename(f)[len("bench_"):]
        open(os.path.join(ROOT, "profiles", "r03_bench_under_rocprof_" + name), "w").write(line[-1] + "\n")
if os.path.exists(os.path.join(out, "sq_config3.json")):
    shutil.copy(os.path.join(out, "sq_config3.json"), os.path.join(ROOT, "profiles", "r03_sq_config3_200steps.json"))
