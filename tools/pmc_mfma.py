"""MFMA utilisation of the two MFMA kernels (k_step, k_eval) from rocprofv3 --pmc passes (own runs, kernel-trace only, as
MI355X_MICROARCH.md prescribes): the derived MfmaUtil metric and the raw f32 MFMA instruction count, turned into TFLOP/s
with the dispatch duration (v_mfma_f32_16x16x4_f32 = 2,048 FLOP per wave instruction).  Run on the GPU box:

    python tools/pmc_mfma.py gpurun_out/pmc_mfma      # writes <dir>/pmc_mfma.json; copy it to profiles/
"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-small-pop", "--steps", "1", "--warmup", "0", "--epochs", "1",
         "--n-train", "2000", "--n-dev", "5600"]
F32_MFMA_PEAK_TFLOPS = 157.3      # 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz (16x16x4 f32: 2,048 FLOP / 32 cycles)


def collect(outdir, tag, counters):
    d = os.path.join(outdir, tag)
    subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", tag, "--"] + BENCH,
                   check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    return list(csv.DictReader(open(f)))


def main():
    outdir = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_mfma")
    os.makedirs(outdir, exist_ok=True)
    res = {}
    passes = {"MfmaUtil": ["MfmaUtil"], "SQ_INSTS_VALU_MFMA_F32": ["SQ_INSTS_VALU_MFMA_F32"]}
    for name, counters in passes.items():
        rows = collect(outdir, name, counters)
        for kern in ("k_step", "k_eval"):
            ks = [r for r in rows if r["Kernel_Name"].startswith("void " + kern) and r["Counter_Name"] == name]
            if not ks:
                continue
            gmax = max(int(r["Grid_Size"]) for r in ks)
            ks = [r for r in ks if int(r["Grid_Size"]) == gmax]
            v = [float(r["Counter_Value"]) for r in ks]
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in ks]
            e = res.setdefault(kern, {"kernel": ks[0]["Kernel_Name"], "grid_threads": gmax})
            e[name] = {"dispatches": len(v), "mean": sum(v) / len(v), "mean_duration_us": 1e6 * sum(dur) / len(dur)}
            if name == "SQ_INSTS_VALU_MFMA_F32":
                tf = [x * 2048.0 / t / 1e12 for x, t in zip(v, dur)]
                e["f32_mfma_tflops"] = sum(tf) / len(tf)
                e["frac_of_f32_mfma_peak"] = e["f32_mfma_tflops"] / F32_MFMA_PEAK_TFLOPS
    res["note"] = ("separate rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --steps 1 --warmup 0 --epochs 1 --n-train 2000 "
                   "--n-dev 5600` (pop 128, conf 4, R=128); durations under counter collection are inflated, so the TFLOP/s here is "
                   "a lower bound; peak = dense f32 MFMA (16x16x4) %.1f TFLOP/s" % F32_MFMA_PEAK_TFLOPS)
    json.dump(res, open(os.path.join(outdir, "pmc_mfma.json"), "w"), indent=1)
    print(json.dumps(res, indent=1)[:1500])


if __name__ == "__main__":
    main()
