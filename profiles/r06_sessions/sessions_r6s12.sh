cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6s12
ntsc-crt_amd/lib/box_speed > gpurun_out/r6s12/box.txt
for d in /sys/class/drm/card*/device; do [ -e $d/pp_dpm_sclk ] && { echo $d; cat $d/pp_dpm_sclk; cat $d/hwmon/hwmon*/power1_cap $d/hwmon/hwmon*/power1_cap_default $d/hwmon/hwmon*/power1_cap_max $d/hwmon/hwmon*/power1_input 2>/dev/null | tr '\n' ' '; echo; }; done >> gpurun_out/r6s12/box.txt 2>&1
bash tools/refresh_profiles.sh r6b > gpurun_out/r6s12/refresh.log 2>&1
timeout 1200 python bench.py --full-json gpurun_out/r6s12/bench_full.json > gpurun_out/r6s12/bench_default.json 2> gpurun_out/r6s12/bench_default.err
tail -c 3500 gpurun_out/r6s12/bench_default.json
ntsc-crt_amd/lib/box_speed >> gpurun_out/r6s12/box.txt
cat gpurun_out/r6s12/box.txt
