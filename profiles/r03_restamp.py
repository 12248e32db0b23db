"""Writes the PMC figures of profiles/r03_traffic_pmc.json (made by r03_collect.py from the r03_profile.sh passes) into
profiles/traffic.json, with the hash of the kernel sources they were measured on.  usage: python3 profiles/r03_restamp.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tp = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tp))
pm = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic_pmc.json")))
for c, k in {"config3": "config3:f64", "config4": "config4:f32", "config2": "config2:f64", "config5": "config5:f64"}.items():
    e, m = t[k], pm[c]
    e["hbm_bytes_per_launch"] = m["hbm_bytes_per_launch"]
    e["hbm_bytes_per_launch_uncorrected"] = m["hbm_bytes_per_launch_uncorrected"]
    e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = m["FETCH_SIZE"]["mean_KiB"], m["WRITE_SIZE"]["mean_KiB"]
    e["sources"], e["sources_sha256_16"] = m["sources"], m["sources_sha256_16"]
    e["source"] = re.sub(r"mean over \d+ / \d+ dispatches \(FETCH [^)]*\)",
                         f"mean over {m['FETCH_SIZE']['dispatches']} / {m['WRITE_SIZE']['dispatches']} dispatches (FETCH "
                         f"{m['FETCH_SIZE']['min_KiB'] / 1024:.1f} ... {m['FETCH_SIZE']['max_KiB'] / 1024:.1f} MiB, WRITE "
                         f"{m['WRITE_SIZE']['min_KiB'] / 1024:.1f} ... {m['WRITE_SIZE']['max_KiB'] / 1024:.1f} MiB)", e["source"])
    print(k, e["hbm_bytes_per_launch"], e["sources_sha256_16"])
json.dump(t, open(tp, "w"), indent=1)
