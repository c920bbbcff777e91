#!/bin/bash
# Everything profiles/<tag>/ holds, from ONE gpurun call at one code state:
#   bash tools/round_profile.sh r06      (on the GPU box, from the repo root; ~15 minutes)
# Output: gpurun_out/profiles_<tag>/ -- copy into profiles/<tag>/.
TAG=${1:-r06}
S=gpurun_out/profiles_$TAG
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/gpu_tests_$TAG.txt
bash tools/profile.sh $TAG > gpurun_out/profile_sh_$TAG.log 2>&1
mv gpurun_out/gpu_tests_$TAG.txt $S/gpu_tests.txt
python bench.py > $S/bench_default_stdout.json 2> $S/bench_default_stderr.txt
STREAMS=4096 bash tools/prof_quick.sh > /dev/null 2>&1; cp gpurun_out/pc/pmc_summary.txt $S/pmc_pipe_per_kernel.txt; rm -rf gpurun_out/pc
bash tools/reg_quick.sh > $S/matcher_phase_split.txt 2>&1
STREAMS=1 bash tools/reg_quick.sh > $S/matcher_phase_split_single_stream.txt 2>&1
bash tools/surf_quick.sh > $S/surface_sort_phase_split.txt 2>&1
bash tools/coral_timing.sh > $S/coral_phase_split.txt 2>&1
python tools/form_sweep.py 2>&1 | grep -v amdgpu.ids > $S/form_sweep.txt
python tools/single_stream.py 2>&1 | grep -v amdgpu.ids > $S/single_stream.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > $S/smoke.txt
timeout 300 python tools/pipe_probe.py 2>&1 | grep -E "^n " > $S/pipe_probe.txt
timeout 300 python tools/verify_quick.py 2>&1 | tail -1 > $S/verify_quick.txt
bash tools/verify_timing.sh > $S/verify_host_timeline.txt 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/calib/cfar_stream_probe.hip -o /tmp/cfar_stream_probe 2>/dev/null && timeout 120 /tmp/cfar_stream_probe > $S/cfar_stream_probe.txt 2>&1
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extras --workload verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('verify', round(d['value']), d['ms_per_step'])"; done > $S/verify_runs.txt
ls -la $S
