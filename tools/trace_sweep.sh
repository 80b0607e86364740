for k in 1 8 3 9 10 11 0 4 12 13; do
  case $k in 1) sk="0 2 8 20 40 57";; 8) sk="0 5 20 40 57";; 3|9|10) sk="0 1 2";; *) sk="0";; esac
  bash tools/trace_phases.sh $k "$sk" SMG_C1_WS=0
done
