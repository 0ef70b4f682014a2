#!/bin/bash
# Diagnostic: where a time step of the cluster LSTM spends its cycles.  Builds a SECOND copy of the library with -DSMIN_LSTM_STAMPS
# (s_memtime at the phase boundaries of workgroup 0) under /tmp and runs tools/lstm_stamps.py against it.   bash tools/lstm_stamps.sh
set -e
ROOT=$(pwd); CS=$ROOT/video-moment-localization_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops \
  -I$ROOT/include -I$CS -DSMIN_LSTM_STAMPS "$@" -c $CS/bilstm_cluster.hip -o /tmp/bilstm_cluster_st.o 2>/dev/null
OBJS=$(ls $CS/*.o | grep -v bilstm_cluster.o | grep -v torch_binding.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/bilstm_cluster_st.o -o /tmp/libsmin_hip_lstm_stamps.so
python3 $ROOT/tools/lstm_stamps.py /tmp/libsmin_hip_lstm_stamps.so
