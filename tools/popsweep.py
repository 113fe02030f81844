import subprocess, sys, json, os
pops = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "8,16,32,64".split(","))]
extra = sys.argv[2:] 
for pop in pops:
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--steps", "1", "--warmup", "1", "--pop", str(pop), "--no-cpu-baseline"] + extra, capture_output=True, text=True)
    ok = False
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            j = json.loads(ln); ok = True
            r = j["roofline"]
            print(pop, round(j["value"], 2), "cand/s", round(j["ms_per_step"], 1), "ms", "sweep", round(r["avg_launch_us"] or 0, 1), "us", round(r["achieved"] or 0), "GB/s", "acc", round(j["config"]["mean_best_dev_acc"], 4), flush=True)
    if not ok: print(pop, "FAILED", out.stderr[-400:])
