# kernel time and rate of the anchored workloads (config 5a Counter, 5b Extract+Count, 5a without N reads)
set -e
B="timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3), "GB/s", round(d["roofline"]["achieved"]))'
$B --workload cfg5a_50M_10k_anchor_m1 | python -c "$J" cfg5a_50M
$B --workload cfg5b_50M_anchor_ec | python -c "$J" cfg5b_50M
$B --workload cfg5a_50M_10k_anchor_m1 --p-n 0 | python -c "$J" cfg5a_50M_noN
