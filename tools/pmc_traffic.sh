#!/bin/bash
# HBM-side traffic only: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no tracing combined with --pmc).
# usage (on the GPU box, from the repo root): DYF_PMC_REGEX=... bash tools/pmc_traffic.sh <outdir> <python script + args...>
set -u
OUT=$1; shift
R=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "${DYF_PMC_REGEX:-.}" --output-format csv -d "$R/$OUT/p$i" -o p$i -- python "$R/$@" > "$R/$OUT/p$i.log" 2>&1
  echo "pass $i rc=$? : $grp"
done
