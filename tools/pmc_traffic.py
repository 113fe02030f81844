"""HBM traffic per launch of the dominant kernel from two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as
MI355X_MICROARCH.md prescribes, plus a calibration of both counters on a kernel with a known byte count
(tools/membench's 3-plane read-modify-write).  Run on the GPU box from the repo root:

    python tools/pmc_traffic.py gpurun_out/pmc        # writes <dir>/pmc_traffic.json; copy it to profiles/

Corrections applied: counter unit KB = 1024 B; FETCH_SIZE x2 on gfx950 (confirmed by the calibration), WRITE_SIZE x1."""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-small-pop", "--steps", "1", "--warmup", "0", "--epochs", "1",
         "--n-train", "2000", "--n-dev", "320"]


def collect(outdir, tag, counter, cmd):
    d = os.path.join(outdir, tag)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", tag, "--"] + cmd,
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]


def avg(rows, pred):
    v = [float(r["Counter_Value"]) for r in rows if pred(r)]
    return sum(v) / max(len(v), 1), len(v)


def main():
    outdir = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
    os.makedirs(outdir, exist_ok=True)
    res = {"raw": {}, "calibration": {}}
    mb = os.path.join(ROOT, "tools", "membench")
    if not os.path.exists(mb):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", mb + ".hip", "-o", mb], check=True)
    true_bytes = 3 * (400 << 20)          # membench k_rmw: 3 planes x 400 MiB read AND written per launch
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = collect(outdir, "cal_" + counter, counter, [mb])
        kb, n = avg(rows, lambda r: "k_rmw" in r["Kernel_Name"])
        res["calibration"][counter] = {"dispatches": n, "avg_raw_KB": kb, "true_bytes": true_bytes, "factor": true_bytes / (kb * 1024.0) if kb else None}
        rows = collect(outdir, "bench_" + counter, counter, BENCH)
        ks = [r for r in rows if r["Kernel_Name"].startswith("void k_step")]
        gmax = max(int(r["Grid_Size"]) for r in ks)
        kb, n = avg(ks, lambda r: int(r["Grid_Size"]) == gmax)      # update+forward launches that co-schedule a chain
        res["raw"][counter] = {"kernel": ks[0]["Kernel_Name"], "dispatches": n, "avg_raw_KB": kb, "grid_threads": gmax}
    rd = res["raw"]["FETCH_SIZE"]["avg_raw_KB"] * 1024.0 * 2.0
    wr = res["raw"]["WRITE_SIZE"]["avg_raw_KB"] * 1024.0
    res["per_launch"] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
    res["note"] = ("separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-cpu-baseline --steps 1 --warmup 0 "
                   "--epochs 1 --n-train 2000 --n-dev 320` (pop 128); FETCH_SIZE x2.00 per MI355X_MICROARCH.md (confirmed by the "
                   "calibration), WRITE_SIZE x1.00; KB = 1024 B")
    json.dump(res, open(os.path.join(outdir, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res["per_launch"]), json.dumps(res["calibration"]))


def small(outdir):
    """Small-population regime (bench.py --workload c2: 16 sampled L=4 confs, R=16, B=20, one epoch of 100 steps): HBM bytes per
    train step of the whole population, persistent resident schedule (k_president, one launch) against launch-per-phase
    (k_step + k_chain), next to the algorithmic 24 B/param + taps of a streaming schedule."""
    outdir = os.path.abspath(outdir)
    os.makedirs(outdir, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--no-cpu-baseline", "--steps", "1", "--warmup", "0",
           "--epochs", "1", "--n-train", "2000", "--n-dev", "320"]
    steps = 100
    res = {}
    for mode, env in (("persistent_resident", {}), ("launch_per_phase", {"MFAS_PERSIST": "0"})):
        os.environ.pop("MFAS_PERSIST", None)
        os.environ.update(env)
        tot = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = collect(outdir, f"small_{mode}_{counter}", counter, cmd)
            ks = [r for r in rows if r["Kernel_Name"].startswith(("void k_president", "void k_persist", "void k_step", "void k_chain"))]
            tot[counter] = sum(float(r["Counter_Value"]) for r in ks) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
            tot["kernels"] = sorted({r["Kernel_Name"].split("(")[0] for r in ks})
            tot["dispatches"] = len(ks)
        res[mode] = {"kernels": tot["kernels"], "dispatches": tot["dispatches"], "read_bytes_per_step": tot["FETCH_SIZE"] / steps,
                     "write_bytes_per_step": tot["WRITE_SIZE"] / steps, "total_bytes_per_step": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps}
    os.environ.pop("MFAS_PERSIST", None)
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; x2.00 / x1.00, KB = 1024 B) summed over the train kernels of "
                   "`bench.py --workload c2 --steps 1 --warmup 0 --epochs 1 --n-train 2000 --n-dev 320` (16 candidates, 100 train steps), "
                   "divided by 100; the first step's parameter load and the last step's store are included")
    json.dump(res, open(os.path.join(outdir, "pmc_traffic_small.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "small":
        small(sys.argv[1])
    else:
        main()
