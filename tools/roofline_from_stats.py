"""Derives the headline roofline figure from the tracked profiler files of one tools/final_profiles.sh run:
algorithmic bytes of the dominant launch / its AverageNs in the one-workload kernel stats / 8 TB/s, and the HBM
traffic from the two --pmc passes (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md).   python tools/roofline_from_stats.py TAG"""
import csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
line = json.loads([l for l in open(os.path.join(root, tag + "_ser.json")) if l.startswith("{")][-1])
roof = line["roofline"]
rows = list(csv.DictReader(open(os.path.join(root, tag + "_ser", "s_kernel_stats.csv"))))
dom = [r for r in rows if r["Name"].startswith("void roi_align_plane_kernel<4, 64>")][0]
avg_ns = float(dom["AverageNs"])
nbytes = roof["algorithmic_bytes_per_launch"]
print("workload: %s" % line["config"]["workload"])
print("kernel: roi_align_plane_kernel<4, 64>   calls %s   AverageNs %.0f (min %s, max %s)" % (dom["Calls"], avg_ns, dom["MinNs"], dom["MaxNs"]))
print("algorithmic bytes per launch: %d" % nbytes)
print("profiler:  %.1f GB/s = %.3f of 8 TB/s" % (nbytes / avg_ns, nbytes / avg_ns / 8000))
print("bench.py events (same run): launch_ms %.4f -> %.1f GB/s = %.3f of 8 TB/s" % (roof["launch_ms"], roof["achieved"], roof["frac"]))
pm = {}
for key, sub, pre in (("WRITE_SIZE", "_pmc_w", "w"), ("FETCH_SIZE", "_pmc_f", "f")):
    vals = []
    for f in glob.glob(os.path.join(root, tag + sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void roi_align_plane_kernel<4, 64>") and r["Counter_Name"] == key:
                vals.append(float(r["Counter_Value"]))
    pm[key] = sum(vals) / len(vals)
traffic = pm["WRITE_SIZE"] * 1024 + 2 * pm["FETCH_SIZE"] * 1024
print("PMC per launch: WRITE_SIZE %.1f KB, FETCH_SIZE %.1f KB (x2 on gfx950) -> HBM traffic %d B = %.3f x algorithmic"
      % (pm["WRITE_SIZE"], pm["FETCH_SIZE"], traffic, traffic / nbytes))
json.dump({"_comment": "HBM bytes per launch of the dominant kernel from two rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE; "
           "tools/final_profiles.sh, tools/prof_kernels.py roi: batch 16, 32000 RoIs, C=512); bench.py copies it into roofline.traffic",
           "kernel": "roi_align_plane_kernel<4, 64>", "WRITE_SIZE_KB": pm["WRITE_SIZE"], "FETCH_SIZE_KB": pm["FETCH_SIZE"],
           "fetch_correction": "x2 on gfx950 (guide: FETCH_SIZE tallies 128-B requests at 64 B)",
           "traffic_bytes_per_launch": int(traffic), "rois": line["config"]["rois_per_step"], "source_tag": tag},
          open(os.path.join(root, tag + "_pmc_traffic.json"), "w"), indent=2)
