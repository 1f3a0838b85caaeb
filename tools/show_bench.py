"""Digest of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("train %.0f tiles/s  %.2f ms/step  conv_igemm frac %.3f  all conv %.3f  other ms %s" % (d["value"], d["ms_per_step"], r["frac"], r["all_conv_launches"]["frac"], r["other_kernels_ms_per_step"]))
e = d.get("extras", {})
if "inference" in e:
    print("inference %.1f MPix/s %.2f ms/frame roofline %s" % (e["inference"]["value"], e["inference"]["ms_per_frame"], json.dumps(e["inference"].get("roofline"))[:600]))
if "f32_path" in e:
    print("f32 path %.0f tiles/s" % e["f32_path"]["value"])
for k, v in e.get("cfg3", {}).items():
    if "error" in v:
        print(k, v)
        continue
    print("%s: %.1f tiles/s %.2f ms/step B=%d params %d  conv frac %.3f whole-step frac %.3f  families %s" % (
        k, v["value"], v["ms_per_step"], v["tiles_per_step"], v["parameters"], v["roofline"]["frac"], v["whole_step"]["frac"], v["roofline"]["families_ms"]))
print("example_json", e.get("example_json"))
print("cpu_baseline", d.get("cpu_baseline"))
