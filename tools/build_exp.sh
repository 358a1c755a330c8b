# A/B builds of the library for experiments: bash tools/build_exp.sh NAME "-DFLAG ..." [source.hip ...]
#   -> tools/bin/libexp_NAME.so   (run with FRCNN_EXP_LIB=tools/bin/libexp_NAME.so python tools/roi5.py ...)
# Only the named sources (default roi_cl.hip) are rebuilt with the flags; the other objects are the product's.
set -e
cd "$(dirname "$0")/../objectdetection_amd/csrc"
N=$1; F=$2; shift 2 || true
SRCS=${@:-roi_cl.hip}
make -j8 >/dev/null
rm -rf /tmp/obj_$N && cp -r ../lib/obj /tmp/obj_$N
for s in $SRCS; do rm -f /tmp/obj_$N/${s%.hip}.o; done
mkdir -p ../../tools/bin
make -j8 OUT=../../tools/bin/libexp_$N.so OBJDIR=/tmp/obj_$N EXTRA="$F" 2>&1 | grep -E "error|warning" || true
ls -la ../../tools/bin/libexp_$N.so
