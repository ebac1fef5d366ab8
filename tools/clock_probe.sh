#!/bin/bash
# tools/clock_probe.sh (GPU box): VERDICT r05 item 6 -- one run, three instruments, one box.
#  1. tools/ubench/clock_probe: s_memtime vs s_memrealtime (constant 100 MHz) on probe CUs and on MFMA-loaded CUs, with the driver's sclk / power sampled meanwhile
#  2. the network's own conv kernels with s_memtime + s_memrealtime stamps (tools/_variants/libmsi_timing.so, built by tools/build_variants.sh timing:"-DMSI_CONV_TIMING")
#  3. 3000 back-to-back network forwards with the driver's sclk / power sampled at ~25 Hz from sysfs (what rocm-smi prints)
cd "$GRAFT_REPO_ROOT" || exit 1
echo "=== 1. clock_probe"; tools/ubench/clock_probe ${1:-1.5}
echo; echo "=== 2. conv kernels of the default fp32 plan (six-product form), batch 1, 320 x 640: stamps of every workgroup"
if [ -f tools/_variants/libmsi_timing.so ]; then
  cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so; cp tools/_variants/libmsi_timing.so matryodshka_amd/libmsi_hip.so
  python tools/conv_timing.py 0 2 4 5 7 8 11 14 16 2>&1 | grep "per block\|launch span" | cut -c1-230
  echo "--- the same at batch 4, 640 x 1280 is not built into conv_timing.py; bf16 plan (configs[2] shapes):"
  python tools/conv_timing.py --bf16 --batch 16 0 4 7 16 2>&1 | grep "per block\|launch span" | cut -c1-230
  cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
fi
echo; echo "=== 3. driver view during 3000 back-to-back forwards (tools/bench_cnn.py), sysfs sampled every 40 ms"
python - <<'PY'
import glob, subprocess, sys, threading, time
freq = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
powr = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
dpm = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
S = {"sclk": [], "power": [], "dpm": []}
stop = False
def sample():
    while not stop:
        try:
            if freq: S["sclk"].append(float(open(freq[0]).read()) / 1e6)
            if powr: S["power"].append(float(open(powr[0]).read()) / 1e6)
            if dpm:
                for l in open(dpm[0]).read().splitlines():
                    if "*" in l: S["dpm"].append(float(l.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
        except Exception as e:
            pass
        time.sleep(0.04)
p = subprocess.Popen([sys.executable, "tools/bench_cnn.py", "--steps", "3000", "--warmup", "20"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
th = threading.Thread(target=sample); th.start()      # (sampled over the whole process: imports and packing are the low-power head, the loop the high-power part)
out = p.communicate()[0]
stop = True; th.join()
print([l for l in out.splitlines() if "cnn forward" in l][-1:])
for k, v in S.items():
    v = sorted(v)
    print("  %-6s %s" % (k, "n/a" if not v else "min %.0f median %.0f p90 %.0f max %.0f (%d samples over the whole process: import + packing + ~5 s of forwards)" % (v[0], v[len(v) // 2], v[int(len(v) * 0.9)], v[-1], len(v))))
PY
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
