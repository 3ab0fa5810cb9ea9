#!/bin/bash
# samples the GPU's clocks / power from sysfs (or rocm-smi) every ~50 ms while a command runs: tools/clock_watch.sh <out> -- <command...>
out=$1; shift; shift
dev=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
hw=$(ls -d $dev/hwmon/hwmon* 2>/dev/null | head -1)
"$@" &
pid=$!
{
  echo "# dev $dev hwmon $hw"
  ls $hw 2>/dev/null | tr '\n' ' '; echo
  while kill -0 $pid 2>/dev/null; do
    s=$(grep '\*' $dev/pp_dpm_sclk 2>/dev/null | tr -d '\n')
    m=$(grep '\*' $dev/pp_dpm_mclk 2>/dev/null | tr -d '\n')
    f1=$(cat $hw/freq1_input 2>/dev/null); f2=$(cat $hw/freq2_input 2>/dev/null)
    p=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null)
    t=$(cat $hw/temp1_input 2>/dev/null)
    echo "$(date +%s.%N | cut -c1-14) sclk[$s] mclk[$m] freq1=$f1 freq2=$f2 power_uW=$p temp=$t"
    sleep 0.05
  done
} > $out 2>&1
wait $pid
