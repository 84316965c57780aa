#!/bin/bash
# Round 6: where the weight-stationary predictor's 78 us go — the same kernel with one ingredient compiled out per build (results of those builds are
# WRONG by construction; only their time is read).  Built here (hipcc cross-compiles), run on the GPU box: tools/tp_lab.py prints observe-alone / step+predictor.
#   base   : the shipped kernel          nocell : no gate nonlinearities / cell update        nomfma : no matrix instructions (operand reads stay)
#   nobar2 : the second workgroup barrier of a timestep dropped (h_t published early: racy)   nocell_nobar2, nomfma_nobar2 : combinations
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = "build" ]; then
  tools/build_variant.sh tp_nocell -DTP_WS_NO_CELL
  tools/build_variant.sh tp_nomfma -DTP_WS_NO_MFMA
  tools/build_variant.sh tp_nobar2 -DTP_WS_NO_BAR2
  tools/build_variant.sh tp_nocell_nomfma -DTP_WS_NO_CELL -DTP_WS_NO_MFMA
  tools/build_variant.sh tp_zerobias -DTP_WS_ZERO_BIAS
  tools/build_variant.sh tp_halfb -DTP_WS_HALF_B
  tools/build_variant.sh tp_zerobias_halfb -DTP_WS_ZERO_BIAS -DTP_WS_HALF_B
  tools/build_variant.sh tp_skeleton_lds_light -DTP_WS_NO_CELL -DTP_WS_NO_MFMA -DTP_WS_ZERO_BIAS -DTP_WS_HALF_B
  exit 0
fi
mkdir -p gpurun_out/r06
timeout 1200 python tools/tp_lab.py --rounds=5 base=multi-uav-pursuit-evasion_amd/libhns.so nocell=build/variants/libhns_tp_nocell.so nomfma=build/variants/libhns_tp_nomfma.so \
  nobar2=build/variants/libhns_tp_nobar2.so nocell_nomfma=build/variants/libhns_tp_nocell_nomfma.so \
  zerobias=build/variants/libhns_tp_zerobias.so halfb=build/variants/libhns_tp_halfb.so zerobias_halfb=build/variants/libhns_tp_zerobias_halfb.so \
  skeleton_lds_light=build/variants/libhns_tp_skeleton_lds_light.so 2>&1 | tee gpurun_out/r06/tp_arms.txt
