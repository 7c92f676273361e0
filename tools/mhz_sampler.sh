#!/bin/bash
# prints the mean "cpu MHz" of CPUs $1..$2 once a second for $3 seconds (who is throttling: clocks or the quota?)
for i in $(seq 1 $3); do awk -v lo=$1 -v hi=$2 '/^processor/{p=$3} /^cpu MHz/{if(p>=lo&&p<=hi){s+=$4;n++}} END{printf "%.0f MHz\n", s/n}' /proc/cpuinfo; sleep 1; done
