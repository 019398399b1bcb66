#!/bin/bash
# Same-box A/B of a kernel variant: builds a second libraglite_hip.so whose kernel source comes from a git revision
# (default HEAD; or another csrc/*.hip as third argument) and leaves it under raglite_amd/_lib/variants/<name>/ -- it travels to the GPU box with the snapshot.
#   bash scripts/ab_variant.sh NAME [REV | --worktree] [SOURCE.hip]          (here, no GPU needed; default source maxsim_stream.hip)
# then on the GPU box:   RAGLITE_HIP_LIB=raglite_amd/_lib/variants/NAME/libraglite_hip.so python scripts/kernel_ab.py 3
set -eu
NAME=$1; REV=${2:-HEAD}; SRC=${3:-maxsim_stream.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
V=$ROOT/raglite_amd/_lib/variants/$NAME
mkdir -p "$V/src"
if [ "$REV" = "--worktree" ]; then
  cp "$ROOT/raglite_amd/csrc/$SRC" "$ROOT/raglite_amd/csrc/common.h" "$V/src/"
else
  git -C "$ROOT" show "$REV:raglite_amd/csrc/$SRC" > "$V/src/$SRC"
  git -C "$ROOT" show "$REV:raglite_amd/csrc/common.h" > "$V/src/common.h"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" -I"$V/src" -c "$V/src/$SRC" -o "$V/variant.o"
OBJS=$(ls "$ROOT"/raglite_amd/_lib/obj/*.o | grep -v $SRC.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$V/variant.o" -o "$V/libraglite_hip.so"
ls -la "$V/libraglite_hip.so"
