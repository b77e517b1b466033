#!/bin/bash
# tools/ab_objs.sh "obj1 obj2 ..." name1="-DFOO=1" name2="..." : library variants that differ in the named objects only (e.g. "hsv_typed_kernels hsv_kernels";
# the other objects are copied from build/), into gst-plugin-rs_amd/build_ab/lib_<name>.so; tools/ab_headline.sh, ab_store_policy.sh, ab_lane_libs.sh,
# exp_colorlut_variants.py measure them on the GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/gst-plugin-rs_amd
AB=$PKG/build_ab
OBJS=$1
shift
make -s -j8 -C "$PKG" >/dev/null
rm -rf "$PKG/build_ab"
mkdir -p "$AB"
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    mkdir -p "$AB/obj_$name"
    cp -p "$PKG"/build/*.o "$PKG"/build/*.hsaco "$PKG"/build/direct_blob.S "$AB/obj_$name/"
    for o in $OBJS; do rm -f "$PKG/build_ab/obj_$name/$o.o"; done
    ( make -s -j2 -C "$PKG" OBJDIR="$AB/obj_$name" OUT="$AB/lib_$name.so" EXTRA_HIPFLAGS="$flags" >/dev/null && rm -rf "$PKG/build_ab/obj_$name" && echo "built $name ($flags)" ) &
done
wait
