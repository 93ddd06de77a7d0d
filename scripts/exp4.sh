set -e
B="timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --p-n 0"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],4), "ms_per_step", round(d["ms_per_step"],4))'
for n in 2048 262144 1048576 2000000 4000000 8000000 16000000; do
$B --miss 0 --reads $n | python -c "$J" m0_cfg3_$n
done
for n in 2048 2000000; do
$B --miss 0 --reads $n --workload cfg2_10M_1k_m0 | python -c "$J" m0_cfg2_1k_$n
done
