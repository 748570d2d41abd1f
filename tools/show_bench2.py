#!/usr/bin/env python3
"""one-screen summary of bench.py JSON lines: python tools/show_bench2.py FILE..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:] if f.endswith(".json") else "")
        continue
    m = d["modes"]
    print("%s: %.0fx  %.2f ms  | %s" % (f, d["value"], d["ms_per_step"], "  ".join("%s enc %.1f dec %.1f" % (k, v["encode_ms"], v["decode_ms"]) for k, v in m.items())))
    p = d.get("parity_f16_vs_exact_teacher_forced")
    if p:
        print("   identical %s  tf: %s disagree, margin %s, logit err %s ok=%s" % (d["config"].get("identical_clips_f16_vs_exact"), p.get("argmax_disagreements"), p.get("max_margin_at_disagreement"), p.get("max_logit_err"), p.get("ok")))
    r = d.get("roofline")
    if r:
        print("   roof %s frac %.3f avg %.4f ms | decode frac %.3f encode frac %.3f | %s" % (r["kernel"], r["frac"], r["avg_launch_ms"], r["phases"]["decode"]["frac"], r["phases"]["encode"]["frac"],
              "  ".join("%s %.1fms" % (k, v["ms"]) + (" %.0fTF" % v["tflops"] if "tflops" in v and k in ("k_gemm", "k_attn_encoder") else "") for k, v in r["kernels"].items())))
    if d.get("value_plugin_path"):
        print("   plugin path %.0fx (%.1f ms)" % (d["value_plugin_path"], d["plugin_path"]["wall_ms"]))
