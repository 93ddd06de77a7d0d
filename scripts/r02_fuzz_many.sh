# one-off: fuzz seeds beyond the 2000 of the suite, through the C ABI against the oracle
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_fuzz; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
: > $out/fuzz_many_${F2Q_FUZZ_TAG:-a}.txt
for lo in ${F2Q_FUZZ_LOS:-2000 5000 8000 11000 14000 17000}; do
  timeout -k 10 500 python tests/fuzz_gpu_many.py $lo $((lo + 3000)) 2>&1 | grep -v amdgpu.ids | tee -a $out/fuzz_many_${F2Q_FUZZ_TAG:-a}.txt
done
