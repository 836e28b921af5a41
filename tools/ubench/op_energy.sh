#!/bin/bash
# Run ON THE GPU BOX: for each instruction class, run tools/ubench/op_energy for a few seconds with a
# rocm-smi power sampler beside it; prints rate, clock and the median package power while it ran.
cd "$(dirname "$0")"
for op in nop fma64 add64 mul64 addu32 cndmask cvt ldsr ldsw; do
  ./op_energy $op 5 > /tmp/oe_$op.txt &
  pid=$!
  sleep 1.5
  p=()
  for i in 1 2 3 4 5; do p+=($(rocm-smi --showpower 2>/dev/null | grep -o "Power (W): [0-9.]*" | grep -o "[0-9.]*$")); sleep 0.5; done
  wait $pid
  med=$(printf "%s\n" "${p[@]}" | sort -n | sed -n 3p)
  echo "$(cat /tmp/oe_$op.txt)  | package power (median of 5) ${med} W"
done
