# A/B builds of the library for experiments: bash tools/build_exp.sh NAME "-DFLAG ..."  -> tools/bin/libexp_NAME.so
# (run with FRCNN_EXP_LIB=tools/bin/libexp_NAME.so python tools/roi5.py ...)
set -e
cd "$(dirname "$0")/../objectdetection_amd/csrc"
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wall \
  -Wno-unused-function $2 -o ../../tools/bin/libexp_$1.so boxes.hip topk.hip nms.hip roi.hip roi_tile.hip \
  roi_backward.hip losses.hip transform.hip rpn_path.hip
