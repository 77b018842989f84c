#!/usr/bin/env python3
"""Copy the artefacts of one `scripts/gpu_round.sh <tag>` visit from gpurun_out/ (scratch) into profiles/ (tracked).
usage: scripts/refresh_profiles.py <tag> [round-prefix, default r03]
Refuses when the live PMC file was not measured on the kernel sources of this checkout."""
import json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
lines = open(os.path.join(out, f"bench_{tag}.json")).read().strip().splitlines()
contract = json.loads(lines[-1])                       # the contract line (<= 6 kB)
line = json.loads([l for l in lines if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])   # the full record
json.dump({"contract_line_bytes": len(lines[-1]), "contract_line": contract}, open(os.path.join(prof, f"{rnd}_bench_contract_line.json"), "w"), indent=1)
pmc = json.load(open(os.path.join(out, "pmc_live.json")))
assert pmc.get("csrc_hash") == bench.csrc_hash(), (pmc.get("csrc_hash"), bench.csrc_hash(), "kernel sources changed since the visit")
shutil.copy(os.path.join(out, "pmc_live.json"), os.path.join(prof, f"pmc_{rnd}.json"))
json.dump({"check_motion": line["edges"]["check_motion"]}, open(os.path.join(prof, f"pmc_{rnd}_edges.json"), "w"), indent=1)
shutil.copy(os.path.join(out, f"prof_{tag}", "summary_all.txt"), os.path.join(prof, f"{rnd}_final_rocprofv3_summary.txt"))
fd = os.path.join(out, f"bench_force_dist_{tag}.json")
if os.path.exists(fd):   # bench.py --gpus 1 --force-dist: one rank through the whole distributed path (its contract line)
    json.dump(json.loads(open(fd).read().strip().splitlines()[-1]), open(os.path.join(prof, f"{rnd}_bench_force_dist.json"), "w"), indent=1)
for src, dst in (("host_latency.json", f"{rnd}_host_latency.json"),
                 ("roadmap_vs_reference_construction.json", f"{rnd}_roadmap_vs_reference_construction.json"),
                 ("mfma_clock_probe.txt", f"{rnd}_mfma_clock_probe.txt"), ("mfma_hazard_probe.txt", f"{rnd}_mfma_hazard_probe.txt"),
                 ("group_test.json", f"{rnd}_group_test.json"), ("group_ranks_2.json", f"{rnd}_group_ranks_2_processes.json"),
                 ("group_ranks_3.json", f"{rnd}_group_ranks_3_processes.json")):
    if os.path.exists(os.path.join(out, src)):
        shutil.copy(os.path.join(out, src), os.path.join(prof, dst))
r = line["roofline"]
print("value", line["value"], "ms/step", line["ms_per_step"], "edges/s", line["value_edges"], "hash", pmc["csrc_hash"])
for k, v in r["binding"]["per_kernel"].items():
    if v["us"] > 50:
        print("  %-36s %8.1f us  valu %.2f  l2 %.2f" % (k, v["us"], v["valu_busy"], v["l2_hit"]))
