ls /sys/class/drm/ | head -30
for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo == $f; cat $f 2>&1 | head -12; done
for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input; do echo == $f; cat $f 2>&1; done
python -c "import amdsmi; print('amdsmi ok')" 2>&1 | tail -1
rocm-smi --showclocks 2>&1 | head -30
python - <<'PY'
import torch, time
print(torch.cuda.get_device_properties(0))
try:
    print("clock_rate", torch.cuda.clock_rate())
except Exception as e:
    print("clock_rate failed", repr(e)[:200])
PY
