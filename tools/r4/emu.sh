cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_emu.txt
{ timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak --mode replicated --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak --mode replicated --streamed --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C4 --mode replicated --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C4 --mode replicated --streamed --no-latency 1 2 4 8; } 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a gpurun_out/r4_emu.txt
