#!/bin/bash
# tools/ab_colorlut.sh name1="-DFOO=1" name2="..." : builds library variants that differ in colorlut_kernels.hip only (the other objects
# are copied from build/), into gst-plugin-rs_amd/build_ab/lib_<name>.so; tools/exp_colorlut_variants.py measures them on the GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/gst-plugin-rs_amd
AB=$PKG/build_ab
make -s -j8 -C "$PKG" >/dev/null
mkdir -p "$AB"
[ "$1" = "--keep" ] && shift || rm -rf "$AB"/*
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    mkdir -p "$AB/$name"
    cp -p "$PKG"/build/*.o "$AB/$name/"
    rm -f "$AB/$name/colorlut_kernels.o"
    ( make -s -C "$PKG" OBJDIR="$AB/$name" OUT="$AB/lib_$name.so" COLORLUT_EXTRA="$flags" >/dev/null && rm -rf "$AB/$name" && echo "built $name ($flags)" ) &
done
wait
