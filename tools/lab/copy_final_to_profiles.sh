#!/bin/bash
# after `gpurun -- 'bash tools/lab/r06_final.sh'`: the batch's files under their profiles/ names (every profile carries the library digest in its header)
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_final
cp gpurun_out/prof_r06/profile.txt profiles/r06_v4_step_kernel.txt
cp gpurun_out/prof_r06_a6t2/profile.txt profiles/r06_step_kernel_a6t2.txt
cp gpurun_out/tp_r06/profile.txt profiles/r06_tp_observe.txt
cp gpurun_out/prof_r06_envgen/profile.txt profiles/r06_envgen_kernels.txt
cp gpurun_out/prof_r06/launch_overlap.txt profiles/r06_launch_overlap.txt
cp $O/bench_final.json profiles/r06_bench_final.json
cp $O/bench_driver_command.json profiles/r06_bench_driver_command.json
cp $O/phase_a6t2.txt profiles/r06_phase_timeline_a6t2.txt
cp $O/phase_small_4096.txt profiles/r06_phase_timeline_small_4096.txt
cp $O/small_batch.txt profiles/r06_small_batch.txt
cp $O/tp_phases.txt profiles/r06_tp_phases.txt
cp $O/tp_widths.txt profiles/r06_tp_widths.txt
cp $O/tp_tiles.txt profiles/r06_tp_tiles.txt
grep -h "sha256" profiles/r06_v4_step_kernel.txt | head -1
cp gpurun_out/prof_r06_a3t2/profile.txt profiles/r06_step_kernel_a3t2.txt 2>/dev/null || true
