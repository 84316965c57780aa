#!/bin/bash
# Round 6, second pass: the same arms on the kernel whose weights are claimed before the timestep loop (csrc/hns_tp.hip: TP_WS_NO_PIN is the round-5 kernel)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06
V=build/variants
timeout 1200 python tools/tp_lab.py --rounds=5 pinned=multi-uav-pursuit-evasion_amd/libhns.so nopin=$V/libhns_tp_nopin.so nocell=$V/libhns_tp_nocell.so nomfma=$V/libhns_tp_nomfma.so \
  nocell_nomfma=$V/libhns_tp_nocell_nomfma.so zerobias_halfb=$V/libhns_tp_zerobias_halfb.so nobar2=$V/libhns_tp_nobar2.so pinned_again=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tee gpurun_out/r06/tp_arms2.txt
