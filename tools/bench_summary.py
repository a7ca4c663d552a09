"""Print the handful of numbers of a bench.py JSON line that the kernel work is steered by."""
import json
import sys

r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
x = r.get("extra", {})
print("entities  value %.3e  us/step %.1f  roofline %.3f" % (r["value"], 1e3 * r["ms_per_step"], r["roofline"]["frac"]))
for k in ("pose_palette", "skinning", "particles", "bodies"):
    if k in x and "roofline" in x[k]:
        rf = x[k]["roofline"]
        print("%-13s %.1f us  frac %.3f" % (k, rf["mean_launch_us"], rf["frac"]))
if "bodies" in x and "broadphase" in x["bodies"]:
    print("broadphase    %.1f us" % (1e3 * x["bodies"]["broadphase"]["ms"]))
if "full_frame" in x:
    print("full frame    %.3f ms (graph %.3f)" % (x["full_frame"]["ms_per_frame"], x["full_frame"].get("ms_per_frame_graph_replay", 0)))
