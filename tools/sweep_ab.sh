#!/bin/bash
# tools/sweep_ab.sh (GPU box): same-box A/B of the sweep variants in tools/_variants (stage times of bench.py --config 2 / 3) + the write-only / copy rates of this box
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import torch, time
for name, n, dt in (("2.5 GB bf16", 1258291200, torch.bfloat16), ("2.5 GB fp32", 629145600, torch.float32)):
    x = torch.empty(n, dtype=dt, device="cuda"); y = torch.empty_like(x)
    for op, f, bytes_ in (("fill (write only)", lambda: x.zero_(), x.numel() * x.element_size()), ("copy (read + write)", lambda: y.copy_(x), 2 * x.numel() * x.element_size())):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%s %-20s %.3f ms  %.2f TB/s" % (name, op, ms, bytes_ / ms / 1e9))
PY
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for r in 1 2; do
for cfg in 2 3; do
  for v in tools/_variants/libmsi_*.so; do
    cp "$v" matryodshka_amd/libmsi_hip.so
    python bench.py --config $cfg --no-cpu-baseline --no-alt-arithmetic --strong-frames 0 --repeats 0 --steps 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $cfg', '$v', j['value'], j['ms_per_step'], {k:v['ms'] for k,v in j['stages'].items() if isinstance(v, dict)})"
  done
done
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
