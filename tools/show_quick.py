"""One line per table of quick-bench outputs: python tools/show_quick.py gpurun_out/qb_a.json ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f)
    rows = [("fbw_rle", d)] + list(d.get("other_workloads", {}).items())
    for name, v in rows:
        ks = " ".join(f"{k['kernel']}:{k['ms_per_step'] * 1e3:.1f}us/{k['tiles']}" for k in v.get("kernels", []))
        print(f"  {name:12s} {v['value']:8.1f} GB/s  {v['ms_per_step'] * 1e3:7.1f} us  frac {v['whole_batch_hbm_frac']:.4f}  | {ks}")
        if "cold_ms_per_step" in v:
            cs = " ".join(f"{k['kernel']}:{k['ms_per_step'] * 1e3:.1f}us" for k in v.get("cold_kernels", []))
            print(f"  {'':12s} cold {v['cold_ms_per_step'] * 1e3:7.1f} us  frac {v['cold_whole_batch_hbm_frac']:.4f}  plan {v.get('plan_ms')} ms  | {cs}")
