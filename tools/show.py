import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        if "detail" in j and "other_kernels_ms" not in j:     # the stdout line is the compact record: the full one is in the side file
            import os
            for base in ("gpurun_out", ".", os.path.dirname(f)):
                p = os.path.join(base, j["detail"])
                if os.path.exists(p):
                    j = json.loads(open(p).read())
                    break
        r = j["roofline"]
        print(f.split("/")[-1], "ms/step %.3f" % j["ms_per_step"], "q/s %.4g" % j["value"], "| main %.3f ms" % r["avg_launch_ms"], "frac %.3f" % r["frac"], j["dtype"],
              "| fb", j["certification_fallback_rows"], "esc", j["escalated_rows"], "err/eps %.3f" % j["rounding_bound_self_check"]["max_err_over_eps"],
              "| fin %.3f" % j["other_kernels_ms"]["finalize_avg"], "fbms %.2f" % j["other_kernels_ms"]["fallback_total"], "|", j.get("check"))
    except Exception as e:
        print(f, "ERR", e)
