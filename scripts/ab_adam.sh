# alternating A/B of the optimizer tail: csrc/optim.hip (default) vs stock torch ops (run through gpurun from the repo root)
set -e
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  for t in 0 1; do
    BENCH_TORCH_ADAM=$t python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/ab/adam${t}_$rep.json 2> gpurun_out/ab/adam${t}_$rep.err
    python -c "import json; d=json.load(open('gpurun_out/ab/adam${t}_$rep.json')); print('torch_adam=$t rep $rep', d['ms_per_step'])"
  done
done
