# run-to-run spread of the default bench line (same box, fresh process each time)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_repeats; mkdir -p $out; : > $out/repeats.txt
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --no-pmc > $out/r$i.json 2> $out/r$i.err
  python -c "import json; d=json.load(open('$out/r$i.json')); print('run $i: value', round(d['value']), 'Mreads/s  ms_per_step', round(d['ms_per_step'],4), ' kernel_ms', round(d['roofline']['kernel_ms'],4), ' frac', round(d['roofline']['frac'],3))" | tee -a $out/repeats.txt
done
