cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s11
ls /sys/class/drm/ > gpurun_out/r6s11/sysfs.txt 2>&1; for d in /sys/class/drm/card*/device; do echo $d >> gpurun_out/r6s11/sysfs.txt; cat $d/pp_dpm_mclk $d/pp_dpm_fclk $d/pp_dpm_sclk >> gpurun_out/r6s11/sysfs.txt 2>&1; ls $d/hwmon/* >> gpurun_out/r6s11/sysfs.txt 2>&1; done
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s11/modes.txt
MODE_ORDERS=-1,1 timeout 900 python tools/mode_probe.py --procs 5 >> gpurun_out/r6s11/modes.txt 2> gpurun_out/r6s11/modes.err
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s11/modes.txt
cut -c1-70,200-900 gpurun_out/r6s11/modes.txt
timeout 600 python tools/ab_sweep.py tools/specs_r6s11.txt --procs 3 > gpurun_out/r6s11/ab.txt 2> gpurun_out/r6s11/ab.err
cat gpurun_out/r6s11/ab.txt
head -40 gpurun_out/r6s11/sysfs.txt
