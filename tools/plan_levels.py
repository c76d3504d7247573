"""Launch list of a benchmark table's plan, per operation (no GPU needed):
python tools/plan_levels.py fbw_rle|dict_string|mixed|wide"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nimble_amd import workloads as WL  # noqa: E402
from nimble_amd.decoder import Plan  # noqa: E402

name = sys.argv[1]
chunks = {"fbw_rle": WL.fbw_rle_table, "dict_string": WL.dict_string_table, "mixed": WL.mixed_table_stripe,
          "wide": WL.wide_table_stripe}[name]()
buf, offs = WL.pack_chunks(chunks)
for per_kernel in (True, False):
    p = Plan(buf, offs, per_kernel=per_kernel)
    print("per operation:" if per_kernel else "fused:")
    for k, es, tiles, level, alg in p.launches():
        print(f"  level {level}  {k:28s} tiles {tiles:7d}  alg MB {alg / 1e6:8.1f}")
    p.close()
