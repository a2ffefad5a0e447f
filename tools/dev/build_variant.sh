#!/bin/bash
# dev-only: build a VARIANT of the HIP library for an A/B run (GARMENTNETS_HIP_LIB=tools/dev/_build/lib_<name>.so): the sources of
# garmentnets_amd/csrc are copied, a patch script (python, reads/writes files in the copy; argument 2) is applied, and the copy is built.
# usage: tools/dev/build_variant.sh <name> <patch.py>
set -e
NAME=$1; PATCH=$2
REPO=$(cd "$(dirname "$0")/../.." && pwd)
DST=$REPO/tools/dev/_build/var_$NAME
rm -rf "$DST"; mkdir -p "$DST/garmentnets_amd" "$DST/include"
cp -r "$REPO/garmentnets_amd/csrc" "$DST/garmentnets_amd/csrc"
cp "$REPO/include/garmentnets_hip.h" "$DST/include/"
( cd "$DST/garmentnets_amd/csrc" && python "$PATCH" )
make -C "$DST/garmentnets_amd/csrc" -j8 > "$DST/build.log" 2>&1 || { tail -30 "$DST/build.log"; exit 1; }
cp "$DST/garmentnets_amd/libgarmentnets_hip.so" "$REPO/tools/dev/_build/lib_$NAME.so"
rm -rf "$DST"
echo "built tools/dev/_build/lib_$NAME.so"
