#!/bin/bash
# Diagnostic: where a round of content_attn_bwd_kernel spends its cycles.  Builds a SECOND copy of the library with
# -DSMIN_ATTN_STAMPS (s_memtime at the phase boundaries, written to a buffer of their own) under /tmp and runs
# tools/attn_stamps.py against it.  The product library is not touched.   bash tools/attn_stamps.sh [extra hipcc flags]
set -e
ROOT=$(pwd); CS=$ROOT/video-moment-localization_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops \
  -I$ROOT/include -I$CS -DSMIN_ATTN_STAMPS "$@" -c $CS/content_attn.hip -o /tmp/content_attn_st.o 2>/dev/null
OBJS=$(ls $CS/*.o | grep -v content_attn.o | grep -v torch_binding.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/content_attn_st.o -o /tmp/libsmin_hip_stamps.so
python3 $ROOT/tools/attn_stamps.py /tmp/libsmin_hip_stamps.so
