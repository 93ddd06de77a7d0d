set -e
B="timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
$B --workload cfg4_50M_100k_m1 | python -c "$J" cfg4_100k
$B --workload cfg2_10M_1k_m0 | python -c "$J" cfg2
$B --workload cfg3_50M_10k_m1 | python -c "$J" cfg3
