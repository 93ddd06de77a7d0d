set -e
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
$B --workload cfg5a_50M_10k_anchor_m1 --reads 5000000 | python -c "$J" cfg5a_5M
$B --workload cfg5b_50M_anchor_ec --reads 5000000 | python -c "$J" cfg5b_5M
