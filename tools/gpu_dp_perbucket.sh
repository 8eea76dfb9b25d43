cd "$GRAFT_REPO_ROOT"
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for rep in 1 2; do
for pb in 4 3 "3,1" 2; do
  echo -n "per-bucket $pb: "; timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --per-bucket $pb 2>&1 | grep "16 channels" | cut -c60-200
done
echo "--- wire at 44 GB/s"
for pb in 4 3 "3,1" 2; do
  echo -n "per-bucket $pb: "; timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --per-bucket $pb --gbs 44 2>&1 | grep "16 channels" | cut -c60-200
done
done
