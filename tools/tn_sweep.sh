export TN_ONLY=1
for cfg in "3 2 4" "3 2 2" "3 1 4" "2 2 4"; do
  set -- $cfg
  echo "cfg nb=$1 spc=$2 spw=$3"
  for o in 0 2 3 4; do HGV_TD_NB=$1 HGV_TD_SPC=$2 HGV_TD_SPW=$3 ONLY=$o timeout -k 10 60 python tools/gemm_probe.py | sed 's/.*| //'; done
done
