"""profiles/<round>_decode_pmc_traffic.json from the two --pmc passes of scripts/pmc_decode.sh: HBM bytes per launch of the decoding
kernels, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM / rocprofv3), WRITE_SIZE exact, unit 1000 B."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", "prof") + "/", os.path.join(ROOT, "profiles") + "/"
KEYS = ("ctc_psi_mfma_kernel<true>", "ctc_state_kernel", "attn_decode_kernel", "attn_fwd_stream_kernel<false, true>", "decode_linear_kernel<2, 1>",
        "decode_linear_kernel<0, 1>", "beam_candidates_topk_kernel", "topk_merge_advance_kernel")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out = {}
    for kind, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = max(glob.glob(G + f"decode_{kind}/*/*counter_collection.csv"), key=os.path.getmtime)
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for k in KEYS:
                if k in r["Kernel_Name"]:
                    agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {})[cn + "_kB_per_launch"] = round(sum(v) / len(v), 1)
            out[k]["launches_counted"] = len(v)
    for d in out.values():
        d["hbm_bytes_per_launch"] = int(round((2 * d.get("FETCH_SIZE_kB_per_launch", 0.0) + d.get("WRITE_SIZE_kB_per_launch", 0.0)) * 1000))
    out["_note"] = ("separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python bench.py --workload asr_decode --steps 1 --warmup 1`, averaged over every "
                    "launch of the two traced batches (all 82 positions: the caches grow with the position); FETCH_SIZE doubled per the gfx950 correction "
                    "(MI355X_MICROARCH.md, HBM / rocprofv3); unit 1000 B")
    json.dump(out, open(f"{P}{tag}_decode_pmc_traffic.json", "w"), indent=1)
    for k, d in out.items():
        if k != "_note":
            print(f"{k}: {d['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch over {d['launches_counted']} launches")


if __name__ == "__main__":
    main()
