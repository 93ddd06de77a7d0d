set -e
B="timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), "Mreads/s kernel_ms", round(d["roofline"]["kernel_ms"],3))'
$B | python -c "$J" nt_m1
F2Q_NT=0 $B | python -c "$J" plain_m1
$B --p-n 0 | python -c "$J" nt_m1_noN
$B --miss 0 --p-n 0 | python -c "$J" nt_m0_noN
$B --miss 0 --p-n 0 --phred 1 | python -c "$J" nt_m0_noN_noPhred
$B --miss 2 --p-n 0 | python -c "$J" nt_m2_noN
