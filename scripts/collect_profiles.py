"""Copies the summaries written by scripts/run_profiles.sh (and the bench lines taken next to it) from gpurun_out/ into profiles/
and prints the figures DESIGN.md section 6 quotes.  Run in the build container after the gpurun call."""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out") + "/", os.path.join(ROOT, "profiles") + "/"


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def main():
    shutil.copy(newest(G + "prof/frontend/*/*kernel_stats.csv"), P + "r01_frontend_bench_kernel_stats.csv")
    shutil.copy(newest(G + "prof/train/*/*kernel_stats.csv"), P + "r01_train_step_kernel_stats.csv")
    shutil.copy(G + "prof/frontend_bench.json", P + "r01_frontend_bench_under_rocprof.json")
    shutil.copy(G + "prof/train_bench.json", P + "r01_train_step_bench_under_rocprof.json")
    for src, dst in (("bench_default.json", "r01_train_step_bench_default.json"), ("bench_frontend_default.json", "r01_frontend_bench_default.json"),
                     ("bench_p512.json", "r01_train_step_bench_prenet512.json"), ("bench_40.json", "r01_train_step_bench_20steps.json")):
        if os.path.exists(G + src):
            shutil.copy(G + src, P + dst)
    out, rows_out = {}, []
    for kind, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(newest(G + f"prof/frontend_{kind}/*/*counter_collection.csv"))):
            n = r["Kernel_Name"]
            name = "frontend_pfa400_kernel" if "pfa400_kernel" in n else "specaug_kernel" if "specaug_kernel" in n else None
            if name:
                agg[name].append(float(r["Counter_Value"]))
                rows_out.append(dict(kernel=name, dispatch=r["Dispatch_Id"], counter=r["Counter_Name"], value=r["Counter_Value"],
                                     start_ns=r["Start_Timestamp"], end_ns=r["End_Timestamp"]))
        for k, v in agg.items():
            out.setdefault(k, {})[cn + "_kB_per_launch"] = sum(v) / len(v)
    for d in out.values():
        # MI355X_MICROARCH.md (HBM / rocprofv3): gfx950 tallies a 16 B/lane stream at half its bytes in FETCH_SIZE -> x2; WRITE_SIZE exact; unit 1000 B
        d["hbm_bytes_per_launch"] = int(round((2 * d["FETCH_SIZE_kB_per_launch"] + d["WRITE_SIZE_kB_per_launch"]) * 1000))
    out["_note"] = ("separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python bench.py --workload frontend --steps 5 --warmup 2`; FETCH_SIZE "
                    "doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3); unit 1000 B")
    json.dump(out, open(P + "r01_frontend_pmc_traffic.json", "w"), indent=1)
    with open(P + "r01_frontend_pmc_dispatches.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows_out[0].keys()))
        w.writeheader()
        w.writerows(rows_out)
    # train step: the same two passes over a few eager steps (scripts/pmc_train.sh), averaged per kernel
    if glob.glob(G + "prof/train_fetch/*/*counter_collection.csv"):
        keys = ("gemm_bf16_nt_dma_kernel<2, 2, 2>", "gemm_bf16_nt_dma_kernel<2, 4, 4>", "gemm_bf16_nt_dma_kernel<2, 2, 1>", "gemm_bf16_kernel<true, true>",
                "attn_fwd_stream_kernel", "attn_bwd_kernel", "adam_step_kernel", "grad_sumsq_kernel", "cast_weights_kernel")
        tr = {}
        for kind, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(newest(G + f"prof/train_{kind}/*/*counter_collection.csv"))):
                for key in keys:
                    if key in r["Kernel_Name"]:
                        agg[key].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                tr.setdefault(k, {})[cn + "_kB_per_launch"] = round(sum(v) / len(v), 1)
                tr[k]["launches_counted"] = len(v)
        for d in tr.values():
            d["hbm_bytes_per_launch"] = int(round((2 * d["FETCH_SIZE_kB_per_launch"] + d["WRITE_SIZE_kB_per_launch"]) * 1000))
        tr["_note"] = ("separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `BENCH_NO_GRAPH=1 python bench.py --steps 3 --warmup 2` (eager steps), averaged over "
                       "every launch of the kernel in the run -- i.e. over the MIX of shapes a kernel serves in the step; FETCH_SIZE doubled per the gfx950 "
                       "correction (MI355X_MICROARCH.md, HBM / rocprofv3); unit 1000 B")
        json.dump(tr, open(P + "r01_train_step_pmc_traffic.json", "w"), indent=1)
    for tag in ("r01_train_step_bench_default", "r01_train_step_bench_20steps", "r01_train_step_bench_prenet512", "r01_train_step_bench_under_rocprof"):
        if os.path.exists(P + tag + ".json"):
            d = json.load(open(P + tag + ".json"))
            r = d["roofline"]
            print(f"{tag}: {d['ms_per_step']} ms/step, {d['value']} utt/s; {r['kernel']}: {r['launches_timed']} launches, {r['avg_launch_ms'] * 1e3:.1f} us, "
                  f"{r['achieved']} TFLOP/s ({r['frac']:.3f}); all GEMMs {r['all_gemm_launches']['achieved']}; named path {r['step_named_path']['achieved']}"
                  + (f"; cpu {d['cpu_baseline']['value']} utt/s ({d['cpu_baseline']['sample'][:40]}...)" if "cpu_baseline" in d else ""))
    for tag in ("r01_frontend_bench_default", "r01_frontend_bench_under_rocprof"):
        d = json.load(open(P + tag + ".json"))
        r = d["roofline"]
        print(f"{tag}: {d['value']} utt/s, {d['ms_per_step']} ms/step; kernel {r['kernel_ms'] * 1e3:.1f} us, {r['achieved']} GB/s ({r['frac']:.3f}), traffic {r['traffic']}"
              + (f"; cpu {d['cpu_baseline']['value']} utt/s" if "cpu_baseline" in d else ""))
    steps = json.load(open(P + "r01_train_step_bench_under_rocprof.json"))["steps"] + json.load(open(P + "r01_train_step_bench_under_rocprof.json"))["warmup"]
    print(f"--- train-step kernel stats per step ({steps} steps traced)")
    for r in csv.DictReader(open(P + "r01_train_step_kernel_stats.csv")):
        if any(k in r["Name"] for k in ("nt_dma_kernel", "gemm_bf16_kernel", "splitk_reduce", "attn_", "layernorm", "ctc_", "featnorm", "frontend_pfa",
                                        "specaug_kernel", "reduce_parts", "cast_weights", "multi_tensor_apply", "row_lse", "ce_", "conv1_", "im2col", "col2im", "bn_", "adam_step", "grad_sumsq",
                                        "sum_partials")):
            print(f"  {float(r['AverageNs']) / 1e3:8.1f} us avg {int(r['Calls']) / steps:6.1f} calls/step {float(r['TotalDurationNs']) / 1e6 / steps:6.3f} ms/step  {r['Name'][:80]}")
    print("--- frontend run")
    for r in csv.DictReader(open(P + "r01_frontend_bench_kernel_stats.csv")):
        if any(k in r["Name"] for k in ("pfa400_kernel", "specaug_kernel")):
            print(f"  {float(r['AverageNs']) / 1e3:.1f} us avg, {r['Calls']} calls  {r['Name'][:70]}")


if __name__ == "__main__":
    main()
