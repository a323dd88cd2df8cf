#!/bin/bash
# Sample GPU power, clocks and temperature (rocm-smi) beside a command.  Usage: tools/power_trace.sh OUT.csv -- command...
OUT=$1; shift; [ "$1" = "--" ] && shift
( while true; do
    echo "$(date +%s.%N) $(rocm-smi -d 0 --showpower --showclocks --showtemp --csv 2>/dev/null | grep card0)"
    sleep 0.2
  done ) > "$OUT" &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER 2>/dev/null
exit $RC
