#!/bin/bash
# random-gather ceilings of the MI355X for the access shapes the matchers use (profiles/gather_bench.hip)
#   here (no GPU):  bash profiles/gather_bench.sh build
#   on the box:     gpurun -- 'bash profiles/gather_bench.sh run'   -> gpurun_out/gather*.jsonl
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p profiles/_bin
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 profiles/gather_bench.hip -o profiles/_bin/gather_bench
else
  timeout -k 10 400 profiles/_bin/gather_bench > gpurun_out/gather.jsonl
  timeout -k 10 400 profiles/_bin/gather_bench 1 stream > gpurun_out/gather_stream.jsonl
fi
