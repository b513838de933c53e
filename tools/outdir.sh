#!/bin/bash
# Per-call output directories (round 6; sourced by every tools/*.sh):
#   OUT=$(new_outdir TAG)   ->  gpurun_out/TAG_<unix time>[_n]/, created, never an existing one
# A retry therefore cannot overwrite the logs of the run that failed before it -- which is how round 4's memory access fault lost
# its only evidence: the scripts then wrote fixed names under gpurun_out/.
new_outdir() {
  local base="gpurun_out/$1_$(date +%s)" d n=0
  d=$base
  while [ -e "$d" ]; do n=$((n + 1)); d="${base}_$n"; done
  mkdir -p "$d"
  echo "$d"
}
