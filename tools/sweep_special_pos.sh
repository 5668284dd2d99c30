for pct in 100 90 80 70 60 50 40; do
  export EZHIP_SPECIAL_PCT=$pct
  echo "PCT=$pct $(python tools/probe_single.py 2>/dev/null | head -1)"
done
unset EZHIP_SPECIAL_PCT
for rb in 4 6; do for pct in 80 70 60; do export EZHIP_SINGLE_RB=$rb EZHIP_SPECIAL_PCT=$pct; echo "RB=$rb PCT=$pct $(python tools/probe_single.py 2>/dev/null | head -1)"; done; done
