for T in 1 2 3 4; do
  echo "== unit target $T"
  python tools/first_update_bench.py --b 1,8 --unit-target $T --label ut$T
  python tools/batch_bench.py --b 1,8 --passes 3 --unit-target $T
done
