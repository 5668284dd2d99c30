for rb in 0 3 4 5 6 7 8 9 10 11 12 13 15; do
  if [ $rb = 0 ]; then unset EZHIP_SINGLE_RB; else export EZHIP_SINGLE_RB=$rb; fi
  echo "RB=$rb $(python tools/probe_single.py 2>/dev/null | tr '\n' ' ')"
done
