cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
{ timeout -k 10 300 python tools/emulate_world.py --config C3 1 2 4 8 && timeout -k 10 200 python tools/emulate_world.py --config C3 --certified --no-latency 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C4 --no-latency 1 2 4 8 && timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak --mode replicated --no-latency 2 4 8; } > gpurun_out/r4_emulated_world.txt 2>&1 || { tail -8 gpurun_out/r4_emulated_world.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r4_emulated_world.txt | cut -c1-330
