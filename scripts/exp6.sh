set -e
B="timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload cfg5a_50M_10k_anchor_m1 --p-n 0"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
for n in 262144 1000000 4000000 16000000; do $B --reads $n | python -c "$J" ms1_$n; done
for n in 1000000 16000000; do $B --reads $n --ms 0 | python -c "$J" ms0_$n; done
for n in 1000000 16000000; do $B --reads $n --miss 0 | python -c "$J" m0_ms1_$n; done
