cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for size in "1080 1920" "540 960"; do set -- $size
echo "base ${2}x${1}: $(SRX_LIBRARY=$PWD/variants/libsrx_base.so timeout -k 10 120 python3 bench.py --workload tvl1 --rows $1 --cols $2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "pairs/s", round(d["ms_per_step"],3), "ms; frac", round(d["roofline"]["frac"],3))')"
done
bash scripts/var_tvl1_rows.sh "2,512,4,512" "2,512,2,512" "2,512,4,256" "2,512,2,256"
