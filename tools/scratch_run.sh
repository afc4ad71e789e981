cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for nb in 1 2 4 7; do
for cfg in "A=1" "DYF_HALO_ROWS=0" "DYF_SPARSE_MIXED=0" "DYF_PAIR_INTERP=0"; do
v=$(env $cfg timeout 300 python bench.py --nb $nb --steps 10 --warmup 3 --no-extra-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['value'])")
echo "nb=$nb $cfg : $v fields/s"
done
done
} > gpurun_out/small_forms.log 2>&1
