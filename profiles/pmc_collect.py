"""Sums rocprofv3 counter_collection CSVs of profiles/pmc_run.sh for the stepper kernel and
prints them per wave-stage (one wave's share of one RK4 stage = one RHS evaluation)."""
import csv, glob, json, sys

out, tag = sys.argv[1], sys.argv[2]
walk = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0   # beams a launched wave walks over (grid = resident workgroups)
tot, kname, ndisp = {}, None, set()
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "crb_step_lean_kernel" not in k and "crb_beam_kernel" not in k:
            continue
        kname = k
        ndisp.add((f, row["Dispatch_Id"]))
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
# (round 2: a wave of the lean stepper walks over several beams, so SQ_WAVES is no longer "beams x waves per beam";
#  profiles/r02_sq_*.json are normalised by beams x waves-per-beam x steps x 4 stages from the bench configuration)
waves = tot.get("SQ_WAVES", 0.0)
stages = 200 * 4
res = {"tag": tag, "kernel": kname,
       "normalisation": f"counter total / (SQ_WAVES x {stages} wave-stages per beam) / {walk:g}: a launched wave of the lean stepper "
                        f"walks over {walk:g} beam(s) at this size (grid = resident workgroups)",
       "counters": tot,
       "per_wave_stage": {k: v / (waves * stages * walk) for k, v in tot.items()} if waves else None}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res["per_wave_stage"], indent=1))
