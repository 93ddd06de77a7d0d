set -e
B="timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --miss 0 --p-n 0"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
for g in 10000 20000 24000 26000 50000 100000 400000; do $B --guides $g | python -c "$J" guides_$g; done
