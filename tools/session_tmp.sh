mkdir -p gpurun_out/s14
ls /sys/class/drm/ | head -30 > gpurun_out/s14/drm.txt
POWER_TRACE_SHORT=1 timeout 200 python tools/power_trace.py 5 > gpurun_out/s14/power_trace.log 2>&1
cat gpurun_out/s14/power_trace.log
