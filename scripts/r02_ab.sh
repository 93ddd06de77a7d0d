cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r02_ab; mkdir -p $out
for rep in 1 2; do
for tree in .old_tree .; do
  cd $GRAFT_REPO_ROOT/$tree
  for w in cfg4_50M_100k_m1 cfg3_2win_50M_10k_m1; do
    timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-extras > $out/x.json 2> $out/x.err
    python -c "import json; d=json.load(open('$out/x.json')); print('$tree', '$w', 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
  done
done
done
