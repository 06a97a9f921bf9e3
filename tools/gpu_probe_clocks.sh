#!/bin/bash
# what this box exposes about clocks / power (round 6): sysfs nodes and the two SMI tools
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d | tr '\n' ' ' | cut -c1-1500; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk current_link_speed gpu_busy_percent mem_busy_percent; do
    [ -r $d/$f ] && { echo "-- $f"; cat $d/$f | head -12; }
  done
  for h in $d/hwmon/hwmon*; do echo "-- $h: $(ls $h | tr '\n' ' ')"; for f in power1_average power1_input power1_cap freq1_input freq2_input temp1_input; do [ -r $h/$f ] && echo "$f $(cat $h/$f)"; done; done
done
echo "== rocm-smi"; timeout 60 rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40
echo "== amd-smi"; timeout 60 amd-smi metric --clock --power 2>&1 | head -60
