set -e
B="timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload cfg5a_50M_10k_anchor_m1"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3), "frac", round(d["roofline"]["frac"],3))'
for g in 1 2 3 4 8 16; do F2Q_AN_GRID=$g $B | python -c "$J" grid_x$g; done
