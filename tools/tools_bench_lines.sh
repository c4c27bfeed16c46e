#!/bin/bash
# The four bench lines a round keeps under profiles/ (run on the GPU box from the repo root):
#   default command, the driver's command, bf16 storage, configs[4].   usage: tools_bench_lines.sh [tag] [first|second|all]
R=${1:-r06}
W=${2:-all}
mkdir -p gpurun_out
if [ "$W" != second ]; then
timeout -k 10 560 python bench.py > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench_line.err
timeout -k 10 560 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_line_driver_command.json 2> gpurun_out/${R}_bench_line_driver_command.err
fi
if [ "$W" != first ]; then
timeout -k 10 560 python bench.py --dtype bf16 > gpurun_out/${R}_bench_line_bf16.json 2> gpurun_out/${R}_bench_line_bf16.err
timeout -k 10 560 python bench.py --config 4 > gpurun_out/${R}_bench_line_config4.json 2> gpurun_out/${R}_bench_line_config4.err
fi
python - <<PY
import json
for f in ["", "_driver_command", "_bf16", "_config4"]:
    try:
        d = json.load(open("gpurun_out/${R}_bench_line%s.json" % f))
    except Exception as e:
        print(f or "default", "unreadable:", e); continue
    vs = d.get("voxel_scatter", {})
    print(f or "default", d["value"], d["ms_per_step"], d["dtype"], round(d["roofline"]["frac"], 4), d["roofline"].get("traffic"),
          vs.get("device_seconds"), vs.get("traffic_over_algorithmic"),
          {k: v.get("scenes_per_s") for k, v in d.get("extras", {}).items() if isinstance(v, dict)})
PY
