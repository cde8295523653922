#!/bin/bash
# Build a variant of librmd_hip.so locally (cross-compile) into build_ab/ for an A/B on the GPU box.
# usage: tools/ab_make.sh <label> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
python - <<PY
import os, shutil
from rpg_open_remode_amd import build
out = build.build_hip(force=True, extra_flags="$2".split(), out=os.path.join("build_ab", "librmd_hip_$1.so"))
PY
echo "built build_ab/librmd_hip_$1.so"
