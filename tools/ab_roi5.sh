# A/B timing of config-5 RoIAlign over experimental builds: bash tools/ab_roi5.sh name1 name2 ...  (GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
echo "product: $(timeout -k 10 120 python $R/tools/roi5.py 20 2>&1 | tail -1)"
for n in "$@"; do
  echo "$n: $(FRCNN_EXP_LIB=$R/tools/bin/libexp_$n.so timeout -k 10 120 python $R/tools/roi5.py 20 2>&1 | tail -1)" || exit 1
done
