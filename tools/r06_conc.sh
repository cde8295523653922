run() { local label=$1; shift; local fails=0; for i in 1 2 3 4 5 6; do env "$@" timeout 200 python -m pytest tests/test_concurrency.py -m gpu -x -q 2>&1 | tail -1 | grep -q failed && fails=$((fails+1)); done; echo "$label: $fails of 6 failed"; }
run default A=1
run engines1 RMD_HIP_COPY_ENGINES=1
run engines2 RMD_HIP_COPY_ENGINES=2
run ring8 RMD_HIP_RING_DEPTH=8
run inplace RMD_HIP_HOST_FRAMES=inplace
run staged RMD_HIP_HOST_FRAMES=staged
