"""Short view of a bench.py line: python tools/show_bench.py gpurun_out/b.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.0f images/s  ms/step %.4f  roofline.frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("stage_ms", {k: round(v, 4) for k, v in d["stage_ms"].items()})
for k in ("secondary_workload", "two_batches_in_flight", "configs3_rank_share_batch32"):
    if k in d:
        print(k, "%.0f images/s" % d[k]["value"], "(median step, GPU clock: %.0f)" % d[k].get("value_median_gpu_clock", 0) if d[k].get("value_median_gpu_clock") else "",
              {a: round(b, 4) for a, b in d[k]["stage_ms"].items()})
for k in ("config5_A200907_k30000_C256", "config5_A805707_k30000_C256", "config5_A200907_k30000_C256_batch64"):
    if k in d:
        c = d[k]
        print(k, "%.0f images/s" % c["images_per_s"], {a: round(b, 4) for a, b in c["stage_ms"].items()},
              "roi_align_hbm_frac %.3f" % c["roi_align_hbm_frac"])
if "next_rows" in d:
    print("next_rows", {k: round(v["ms"], 4) for k, v in d["next_rows"].items() if isinstance(v, dict) and "ms" in v})
if d.get("cpu_baseline"):
    print("cpu_baseline %.2f images/s on %d threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
