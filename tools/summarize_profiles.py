"""Turn one tools/profile_round.sh pass (gpurun_out/<tag>/) into the files committed under profiles/:
  <tag>_bench_n1.json, <tag>_<workload>_kernel_stats.csv, <tag>_<workload>_pmc_avg_per_launch.json, hbm_traffic.json
HBM bytes follow the gfx950 rule of /opt/skills/guides/MI355X_MICROARCH.md: (2*FETCH_SIZE + WRITE_SIZE) KiB.
usage: python tools/summarize_profiles.py r01 [workload-name]"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload = sys.argv[2] if len(sys.argv) > 2 else "ringvrf_ring1024_batch1024"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)

line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
json.loads(line)
with open(f"profiles/{tag}_bench_n1.json", "w") as f:
    f.write(line + "\n")

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_{workload}_kernel_stats.csv")

def short(name):
    return name.split("(")[0]


# the --stats summary averages every launch of the process, set-up included (one 60 ms launch builds the by-parts
# bases); the bench line's figure comes from its second pass (the same K steps with the per-kernel timers on).  From the
# kernel trace: the last 2 * per_step accumulate launches = that pass of the profiled run (profile_round.sh uses --steps 2).
trace = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if r["Kernel_Name"].split("(")[0].endswith("k_g1_accumulate")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per_step = int(round(json.loads(line)["roofline"].get("launches_per_step", 6)))
    timed = rows[-2 * per_step:]
    if timed:
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in timed]
        with open(f"profiles/{tag}_{workload}_accumulate_timed_region.json", "w") as f:
            json.dump({"_note": "k_g1_accumulate launches of the 2 timed steps of the rocprofv3 --kernel-trace run (set-up and warm-up "
                                "launches excluded); compare with roofline.avg_kernel_ms of the bench line",
                       "launches": len(dur), "avg_ms": sum(dur) / len(dur), "per_launch_ms": [round(x, 3) for x in dur]}, f, indent=1)

sums = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k][row["Counter_Name"]] += 1
# the clock the chip held under the bucket walk in the --pmc pass: GRBM_GUI_ACTIVE counts cycles of all 8 XCDs
clock_cycles = clock_ns = 0.0
for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and short(row["Kernel_Name"]).endswith("k_g1_accumulate") and int(row["Grid_Size"]) >= 2000000:
                clock_cycles += float(row["Counter_Value"]) / 8.0
                clock_ns += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
avg = {k: {c: sums[k][c] / launches[k][c] for c in sorted(sums[k])} for k in sums}
for k in avg:
    avg[k]["launches_per_pass"] = max(launches[k].values())
with open(f"profiles/{tag}_{workload}_pmc_avg_per_launch.json", "w") as f:
    json.dump(avg, f, indent=1)

acc = avg.get("dr::k_g1_accumulate")
if acc and "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    traffic = {
        "_note": "HBM bytes per k_g1_accumulate launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 correction; average over the launches of one bench step",
        "_source": f"profiles/hbm_traffic.json: rocprofv3 --pmc passes of tools/profile_round.sh {tag} (a separate run of the same workload, "
                   "NOT measured inside bench.py)",
        workload: {
            "k_g1_accumulate_bytes_per_launch": (2 * acc["FETCH_SIZE"] + acc["WRITE_SIZE"]) * 1024,
            "FETCH_SIZE_KB": acc["FETCH_SIZE"],
            "WRITE_SIZE_KB": acc["WRITE_SIZE"],
        },
    }
    if clock_ns:
        traffic["_clock_ghz"] = round(clock_cycles / clock_ns, 4)
        traffic["_clock_source"] = (f"GRBM_GUI_ACTIVE / 8 XCDs / launch duration over the dense k_g1_accumulate launches of the rocprofv3 --pmc pass of "
                                    f"tools/profile_round.sh {tag} (the profiled run; bench.py itself is not under the profiler)")
    with open("profiles/hbm_traffic.json", "w") as f:
        json.dump(traffic, f, indent=1)
print("wrote profiles for", tag, "; kernels with counters:", len(avg))
