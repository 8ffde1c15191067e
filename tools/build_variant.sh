#!/bin/bash
# Build a variant of the library into variants/<name>.so: only csrc/ltr_scorer.hip is recompiled with the given -D flags,
# the other translation units are compiled once into $OBJ/ and reused.   tools/build_variant.sh <name> [-Dflags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
CS=nn-with-pytorch-personalized-losses_amd/csrc
# EXTRA_ALL="<flags>" applies flags to EVERY translation unit (objects then live in variants/obj_<name>/)
OBJ=variants/obj
[ -n "$EXTRA_ALL" ] && OBJ=variants/obj_$NAME
mkdir -p $OBJ
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize $EXTRA_ALL"
for f in ltr_data ltr_encoder ltr_losses ltr_metrics ltr_risk; do
  if [ ! -f $OBJ/$f.o ] || [ $CS/$f.hip -nt $OBJ/$f.o ] || [ -n "$(find $CS include -name '*.h' -newer $OBJ/$f.o)" ]; then
    hipcc $FLAGS -c $CS/$f.hip -o $OBJ/$f.o &
  fi
done
hipcc $FLAGS "$@" -c $CS/ltr_scorer.hip -o $OBJ/scorer_$NAME.o
wait
hipcc -shared -fPIC --offload-arch=gfx950 $OBJ/ltr_data.o $OBJ/ltr_encoder.o $OBJ/ltr_losses.o $OBJ/ltr_metrics.o $OBJ/ltr_risk.o $OBJ/scorer_$NAME.o -o variants/$NAME.so
ls -la variants/$NAME.so
