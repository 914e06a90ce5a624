"""Copies the summaries written by scripts/run_profiles.sh from gpurun_out/prof/ into profiles/<round>_* and prints the figures DESIGN.md
quotes.  Run in the build container after the gpurun call:   python scripts/collect_profiles.py r03"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", "prof") + "/", os.path.join(ROOT, "profiles") + "/"

TRAIN_KERNELS = ("gemm_bf16_nt_dma_kernel<4, 2, 1, 0, false>", "gemm_bf16_nt_dma_kernel<4, 2, 1, 2, false>", "gemm_bf16_nt_dma_kernel<2, 2, 2, 2, false>", "gemm_bf16_nt_dma_kernel<2, 4, 4, 0, false>", "gemm_bf16_nt_dma_kernel<2, 2, 1, 2, false>",
                 "gemm_bf16_nt_dma_kernel<2, 1, 1, 1, false>", "wgrad_grouped_kernel<2, 4, 4>", "gemm_bf16_kernel<true, true>", "attn_fwd_stream_kernel",
                 "attn_bwd_kl_kernel<8, true>", "attn_bwd_kl_kernel<4, true>", "adam_step_kernel", "grad_sumsq_kernel", "cast_weights_kernel", "layernorm_bwd_vec_kernel", "layernorm_fwd_vec_kernel")


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


def traffic(prefix, keys, note):
    """{kernel: FETCH/WRITE kB per launch, HBM bytes per launch} from the two --pmc passes gpurun_out/prof/<prefix>_{fetch,write}."""
    out, rows = {}, []
    for kind, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(newest(G + f"{prefix}_{kind}/*/*counter_collection.csv"))):
            for key in keys:
                if key.rstrip(">") in r["Kernel_Name"]:  # (a key names a kernel up to template parameters added after it was written)
                    agg[key].append(float(r["Counter_Value"]))
                    rows.append(dict(kernel=key, dispatch=r["Dispatch_Id"], counter=r["Counter_Name"], value=r["Counter_Value"],
                                     start_ns=r["Start_Timestamp"], end_ns=r["End_Timestamp"]))
        for k, v in agg.items():
            out.setdefault(k, {})[cn + "_kB_per_launch"] = round(sum(v) / len(v), 1)
            out[k]["launches_counted"] = len(v)
    for d in out.values():
        # MI355X_MICROARCH.md (HBM / rocprofv3): gfx950 tallies a 16 B/lane stream at half its bytes in FETCH_SIZE -> x2; WRITE_SIZE exact; unit 1000 B
        d["hbm_bytes_per_launch"] = int(round((2 * d.get("FETCH_SIZE_kB_per_launch", 0.0) + d.get("WRITE_SIZE_kB_per_launch", 0.0)) * 1000))
    out["_note"] = note
    return out, rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    for src, dst in (("bench_default.json", "train_step_bench_default"), ("bench_p512.json", "train_step_bench_prenet512"),
                     ("bench_frontend.json", "frontend_bench_default"), ("bench_tts.json", "tts_step_bench_default"),
                     ("train_bench.json", "train_step_bench_under_rocprof"), ("frontend_bench.json", "frontend_bench_under_rocprof"),
                     ("tts_bench.json", "tts_step_bench_under_rocprof"), ("bench_decode.json", "decode_bench_default"),
                     ("decode_bench.json", "decode_bench_under_rocprof")):
        shutil.copy(G + src, f"{P}{tag}_{dst}.json")
    for d, dst in (("train", "train_step_kernel_stats"), ("frontend", "frontend_bench_kernel_stats"), ("tts", "tts_step_kernel_stats"),
                   ("decode", "decode_kernel_stats")):
        shutil.copy(newest(G + f"{d}/*/*kernel_stats.csv"), f"{P}{tag}_{dst}.csv")
    fe, rows = traffic("frontend", ("frontend_pfa400_kernel", "specaug_kernel"),
                       "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python bench.py --workload frontend --steps 5 --warmup 2`; FETCH_SIZE doubled per "
                       "the gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3); unit 1000 B")
    json.dump(fe, open(f"{P}{tag}_frontend_pmc_traffic.json", "w"), indent=1)
    with open(f"{P}{tag}_frontend_pmc_dispatches.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    tr, _ = traffic("train", TRAIN_KERNELS,
                    "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `BENCH_NO_GRAPH=1 BENCH_NO_PREFETCH=1 python bench.py --steps 3 --warmup 2` (eager steps: every "
                    "dispatch is counted), averaged over every launch of the kernel in the run -- i.e. over the MIX of shapes a kernel serves in the step; FETCH_SIZE "
                    "doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3); unit 1000 B")
    json.dump(tr, open(f"{P}{tag}_train_step_pmc_traffic.json", "w"), indent=1)
    # matrix-core busy fraction per kernel from the SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass: the counter sums busy cycles over the
    # 1024 SIMDs of the chip (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md), GRBM_GUI_ACTIVE the active cycles of the 8 XCDs
    try:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(newest(G + "train_mfma/*/*counter_collection.csv"))):
            for key in TRAIN_KERNELS:
                if key.rstrip(">") in r["Kernel_Name"]:
                    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        mf = {}
        for k, c in agg.items():
            busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
            if busy and act:
                mf[k] = dict(launches_counted=len(busy), mfma_busy_cycles_per_launch=round(sum(busy) / len(busy)),
                             gui_active_cycles_per_launch=round(sum(act) / len(act)),
                             mfma_busy_frac=round(sum(busy) / (128.0 * sum(act)), 4))
        mf["_note"] = ("`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass of `BENCH_NO_GRAPH=1 BENCH_NO_PREFETCH=1 python bench.py --steps 3 --warmup 2`; "
                       "mfma_busy_frac = busy cycles summed over the chip's 1024 SIMDs / (1024 x GRBM_GUI_ACTIVE / 8), averaged over the launches of a kernel")
        json.dump(mf, open(f"{P}{tag}_train_step_pmc_mfma.json", "w"), indent=1)
        for k in TRAIN_KERNELS:
            if k in mf:
                print(f"  mfma busy {k}: {mf[k]['mfma_busy_frac']:.3f} over {mf[k]['launches_counted']} launches")
    except (ValueError, OSError) as e:
        print("no MFMA pass:", e)
    for name in ("train_step_bench_default", "train_step_bench_prenet512", "train_step_bench_under_rocprof", "tts_step_bench_default", "tts_step_bench_under_rocprof"):
        d = json.load(open(f"{P}{tag}_{name}.json"))
        r = d["roofline"]
        print(f"{tag}_{name}: {d['ms_per_step']} ms/step, {d['value']} utt/s; {r['kernel']}: {r['launches_timed']} launches, {r['avg_launch_ms'] * 1e3:.1f} us, "
              f"{r['achieved']} TFLOP/s ({r['frac']:.3f}), traffic {r['traffic']}; all GEMMs {r['all_gemm_launches']['achieved']}"
              + (f"; named path {r['step_named_path']['achieved']}" if r.get("step_named_path") else "")
              + (f"; cpu {d['cpu_baseline']['value']} utt/s ({d['cpu_baseline']['sample'][:50]}...)" if "cpu_baseline" in d else ""))
    for name in ("frontend_bench_default", "frontend_bench_under_rocprof"):
        d = json.load(open(f"{P}{tag}_{name}.json"))
        r = d["roofline"]
        print(f"{tag}_{name}: {d['value']} utt/s, {d['ms_per_step']} ms/step; kernel {r['kernel_ms'] * 1e3:.1f} us, {r['achieved']} GB/s ({r['frac']:.3f}), traffic {r['traffic']}"
              + (f"; cpu {d['cpu_baseline']['value']} utt/s" if "cpu_baseline" in d else ""))
    d = json.load(open(f"{P}{tag}_decode_bench_default.json"))
    r = d["roofline"]
    print(f"{tag}_decode_bench_default: {d['ms_per_step']} ms per 32-utterance batch, {d['value']} utt/s; {r['kernel']} {r['kernel_ms'] * 1e3:.1f} us, {r['achieved']} GB/s; "
          f"{r['note']}" + (f"; cpu {d['cpu_baseline']['value']} utt/s ({d['cpu_baseline']['sample'][:80]}...)" if "cpu_baseline" in d else ""))
    for wl, extra in (("train_step", 1), ("tts_step", 1), ("decode", 1)):
        b = json.load(open(f"{P}{tag}_{wl}_bench_under_rocprof.json" if wl != "decode" else f"{P}{tag}_decode_bench_under_rocprof.json"))
        steps = b["steps"] + b["warmup"] + extra  # (+ the instrumented eager step after the timed region)
        rows = list(csv.DictReader(open(f"{P}{tag}_{wl}_kernel_stats.csv")))  # (decode: a "step" is one 32-utterance batch)
        print(f"--- {wl} kernel stats per step ({steps} steps traced; total {sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps:.2f} ms of kernels, "
              f"{sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step)")
        for r in rows[:28]:
            print(f"  {float(r['AverageNs']) / 1e3:8.1f} us avg {int(r['Calls']) / steps:6.1f} calls/step {float(r['TotalDurationNs']) / 1e6 / steps:6.3f} ms/step  {r['Name'][:90]}")
    print("--- frontend run")
    for r in csv.DictReader(open(f"{P}{tag}_frontend_bench_kernel_stats.csv")):
        if any(k in r["Name"] for k in ("pfa400_kernel", "specaug_kernel")):
            print(f"  {float(r['AverageNs']) / 1e3:.1f} us avg, {r['Calls']} calls  {r['Name'][:70]}")
    for k in TRAIN_KERNELS:
        if k in tr:
            print(f"  traffic {k}: {tr[k]['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch over {tr[k]['launches_counted']} launches")


if __name__ == "__main__":
    main()
