# alternating A/B of the split-K workgroup target of the weight-gradient GEMMs (run through gpurun from the repo root)
set -e
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for t in 256 128 192 320; do
    SC_WGRAD_WG_TARGET=$t python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/ab/w${t}_$rep.json 2> gpurun_out/ab/w${t}_$rep.err
    python -c "import json; d=json.load(open('gpurun_out/ab/w${t}_$rep.json')); print('target', $t, 'rep', $rep, d['ms_per_step'])"
  done
done
