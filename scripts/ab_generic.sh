# alternating A/B of the train step between two git-stashed library states is awkward on the box; this script instead toggles an env knob:
#   bash scripts/ab_generic.sh VAR A B   (run through gpurun from the repo root)
set -e
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  for val in $2 $3; do
    env $1=$val python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/ab/g_${val}_$rep.json 2> gpurun_out/ab/g_${val}_$rep.err
    python -c "import json; d=json.load(open('gpurun_out/ab/g_${val}_$rep.json')); print('$1=$val rep $rep', d['ms_per_step'])"
  done
done
