cd "$(dirname "$0")/../.."
for v in default "$@"; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  echo "== $v"; timeout 200 python tools/dev/with_lib.py $L tools/dev/check_w4f.py 16 2>&1 | grep "Cin=" | head -2
done
