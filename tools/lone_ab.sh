for c in w8r2 w16 w8 w4r4 w16m2; do
  echo "== $c"; TWL_FAST_CFG=$c timeout 120 python tools/quick_bench.py 128 10000 prof 2>&1 | tail -1
done
