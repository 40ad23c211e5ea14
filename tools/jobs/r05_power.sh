# is the step power-limited?  samples the card's power / shader clock (hwmon / rocm-smi) while bench.py runs
set -u
R=$PWD; OUT=$R/gpurun_out
python3 -c "from vpd_amd.boxid import gpu_unique_id; print('gpu_unique_id', gpu_unique_id(0))" > $OUT/r05_power.txt 2>&1
ls /sys/class/drm/ >> $OUT/r05_power.txt 2>&1
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h | tr '\n' ' '; echo; cat $h/power1_cap 2>/dev/null; cat $h/power1_cap_max 2>/dev/null; done >> $OUT/r05_power.txt 2>&1
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 >> $OUT/r05_power.txt
python3 bench.py --no-cpu-baseline --no-apply --repeats 3 --steps 300 --warmup 20 --profile-steps 0 > $OUT/r05_power_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do
  for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "sample $i $h power $(cat $h/power1_input 2>/dev/null) uW cap $(cat $h/power1_cap 2>/dev/null) sclk $(cat $h/freq1_input 2>/dev/null) Hz temp $(cat $h/temp2_input 2>/dev/null)"; done
  sleep 0.5
done >> $OUT/r05_power.txt 2>&1
wait $BP
python3 -c "
import json; d=json.loads(open('$OUT/r05_power_bench.json').read().strip().splitlines()[-1]); print('bench %.1f crops/s %.3f ms' % (d['value'], d['ms_per_step']))" >> $OUT/r05_power.txt
grep 'sample' $OUT/r05_power.txt | sort -t' ' -k5 -n -r | head -12; tail -1 $OUT/r05_power.txt
