#!/bin/bash
# round 6, the verification batch: whole GPU suite, smoke(), the default bench line and the driver's command, and every profile of profiles/r06_* from
# THIS build (tools/make_profile_txt.py refuses summaries older than libhns.so and heads each file with the library's digest)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python tools/bench_line.py < $O/bench_final.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "driver-command bench rc=$?"
python tools/bench_line.py < $O/bench_driver_command.json | head -3
bash tools/profile_step.sh r06 > $O/profile_step.log 2>&1; tail -14 $O/profile_step.log
KERNEL=hns_step_v4_kernelILi6ELi2 BYTES=208207872 bash tools/profile_step.sh r06_a6t2 --agents 6 --targets 2 --cylinders 16 > $O/profile_a6t2.log 2>&1; tail -12 $O/profile_a6t2.log
bash tools/profile_tp.sh r06 > $O/profile_tp.log 2>&1; tail -10 $O/profile_tp.log
bash tools/profile_envgen.sh r06_envgen > $O/profile_envgen.log 2>&1; tail -12 $O/profile_envgen.log
timeout 300 python tools/tp_phases.py 65536 2>&1 | grep -v amdgpu.ids > $O/tp_phases.txt; tail -3 $O/tp_phases.txt
timeout 600 python tools/tp_widths.py 65536 2>&1 | grep -v amdgpu.ids > $O/tp_widths.txt; cat $O/tp_widths.txt
timeout 300 python tools/small_batch.py 2048 4096 8192 16384 2>&1 | grep -v amdgpu.ids > $O/small_batch.txt; cat $O/small_batch.txt
timeout 300 python tools/phase_profile.py --envs=4096 --cylinders=5 --mapping=small --waves 2>&1 | grep -v amdgpu.ids > $O/phase_small_4096.txt; head -16 $O/phase_small_4096.txt
timeout 300 python tools/phase_profile.py --envs=65536 --agents=6 --targets=2 --cylinders=16 2>&1 | grep -v amdgpu.ids > $O/phase_a6t2.txt; head -18 $O/phase_a6t2.txt
timeout 600 python tools/tp_tiles.py 2>&1 | grep -v amdgpu.ids > $O/tp_tiles.txt; cat $O/tp_tiles.txt
KERNEL=hns_step_v4_kernelILi3ELi2 BYTES=105971712 bash tools/profile_step.sh r06_a3t2 --agents 3 --targets 2 --cylinders 8 > $O/profile_a3t2.log 2>&1; tail -12 $O/profile_a3t2.log
