"""L2 memory-side REQUEST counters of the headline kernel: how the fabric bytes FETCH_SIZE / WRITE_SIZE report
split into requests by size and destination — TCC_EA0_RDREQ (all) / _32B / _DRAM, TCC_EA0_WRREQ (all) / _64B / _DRAM, and the L2's own
hit / miss counts — per launch of k_step, separate rocprofv3 --pmc passes (one counter group each).  The stack exposes no Infinity-Cache
(MALL) hit counter: "destined for DRAM (MC)" is counted in front of the memory-side cache.
usage (GPU box, repo root): python tools/pmc_requests.py gpurun_out/pmc_req     -> <dir>/pmc_requests.json"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-small-pop", "--steps", "1", "--warmup", "0", "--epochs", "1",
         "--n-train", "2000", "--n-dev", "320"]
GROUPS = [["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum"],
          ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"]]


def main():
    outdir = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_req")
    os.makedirs(outdir, exist_ok=True)
    res = {"per_launch": {}, "errors": {}}
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(outdir, f"g{gi}")
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(["rocprofv3", "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", f"g{gi}", "--"] + BENCH,
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not fs:
            res["errors"][",".join(grp)] = (r.stderr or "")[-400:]
            continue
        rows = [x for x in csv.DictReader(open(fs[0])) if x["Kernel_Name"].startswith("void k_step")]
        if not rows:
            continue
        gmax = max(int(x["Grid_Size"]) for x in rows)
        for c in grp:
            v = [float(x["Counter_Value"]) for x in rows if x["Counter_Name"] == c and int(x["Grid_Size"]) == gmax]
            if v:
                res["per_launch"][c] = {"avg": sum(v) / len(v), "dispatches": len(v)}
    p = res["per_launch"]
    g = lambda k: p.get(k, {}).get("avg")
    if g("TCC_EA0_RDREQ_sum"):
        rd, r32, rdd = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum") or 0.0, g("TCC_EA0_RDREQ_DRAM_sum")
        res["derived_read"] = {"requests": rd, "frac_32B": r32 / rd, "frac_destined_for_DRAM": (rdd / rd) if rdd is not None else None,
                               "bytes_if_64B_each": (rd - r32) * 64 + r32 * 32, "bytes_if_128B_each": (rd - r32) * 128 + r32 * 32}
    if g("TCC_EA0_WRREQ_sum"):
        wr, w64, wrd = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_64B_sum") or 0.0, g("TCC_EA0_WRREQ_DRAM_sum")
        res["derived_write"] = {"requests": wr, "frac_64B": w64 / wr, "frac_destined_for_DRAM": (wrd / wr) if wrd is not None else None,
                                "bytes": w64 * 64 + (wr - w64) * 32}
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
        res["derived_l2"] = {"hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))}
    res["note"] = ("separate rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-small-pop --steps 1 --warmup 0 --epochs 1 --n-train 2000 --n-dev 320` "
                   "(pop 128, conf 4, R=128), the k_step launches of the largest grid (update + forward sweeps that co-schedule a chain); raw counter values")
    json.dump(res, open(os.path.join(outdir, "pmc_requests.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "note"})[:1500])


if __name__ == "__main__":
    main()
