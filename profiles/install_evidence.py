#!/usr/bin/env python3
"""Copies an evidence run (profiles/scripts/r3_evidence2.sh on the GPU box, merged back into gpurun_out/) into the
tracked profiles/ tree: rocprofv3 summaries + kernel stats per workload, bench lines, traffic.json.
usage: python profiles/install_evidence.py [run tag prefix, default r3f] [round tag, default r03]"""
import glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = sys.argv[1] if len(sys.argv) > 1 else "r3f"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
names = {"res2a_u8": "res2a-u8", "res2a_s32": "res2a-s32", "vgg_f32": "vgg-f32", "res3_u8": "res3-u8",
         "res4_u8": "res4-u8", "res5_u8": "res5-u8"}
tj = json.load(open(os.path.join(P, "traffic.json")))
for short, key in names.items():
    d = os.path.join(G, "pmc_%s_%s" % (run, short))
    summ = os.path.join(d, "summary.txt")
    if not os.path.exists(summ):
        print("skip", short)
        continue
    shutil.copy(summ, os.path.join(P, "%s_%s_rocprofv3_summary.txt" % (rnd, short)))
    # the python process's kernel stats: the csv with the most calls of a dfx kernel
    best, best_calls = None, -1
    for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
        for line in open(f):
            m = re.match(r'"void dfx::[^"]*",(\d+),', line)
            if m and int(m.group(1)) > best_calls:
                best, best_calls = f, int(m.group(1))
    if best:
        shutil.copy(best, os.path.join(P, "%s_%s_kernel_stats.csv" % (rnd, short)))
    vals, blocks = {}, 0
    for line in open(summ):  # the first "# PMC per dispatch" block = the workload's own kernel
        if line.startswith("# PMC per dispatch"):
            blocks += 1
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
        if m and blocks == 1:
            vals[m.group(1)] = float(m.group(2))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        tj[key] = int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024))
        tj["_raw"][key] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"]}
        print(short, "calls", best_calls, "traffic", tj[key])
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
fin = os.path.join(P, rnd, "final")
os.makedirs(fin, exist_ok=True)
for f in glob.glob(os.path.join(G, run, "bench_*.json")) + [os.path.join(G, run, "general.txt"),
                                                            os.path.join(G, run, "pytest_gpu.log")]:
    if os.path.exists(f):
        shutil.copy(f, fin)
log = os.path.join(G, run + "_evidence.log")
if os.path.exists(log):
    shutil.copy(log, os.path.join(fin, "evidence_run.log"))
for src, dst in (("bench_default.json", "bench_%s_res2a_s32.json"), ("bench_driver_style.json", "bench_%s_res2a_s32_driver_style.json"),
                 ("bench_res2a_u8.json", "bench_%s_res2a_u8.json"), ("bench_vgg_f32.json", "bench_%s_vgg_f32.json"),
                 ("bench_concat.json", "bench_%s_concat.json")):
    f = os.path.join(G, run, src)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, dst % rnd))
