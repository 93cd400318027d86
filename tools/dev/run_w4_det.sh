cd "$(dirname "$0")/../.."
for v in "$@"; do
  echo "== $v"; bash tools/dev/test_variant.sh $v winograd4 2>&1 | tail -2
done
