#!/bin/bash
# Builds a second copy of the product tree under tools/_build/<name> with extra defines for kernels.hip, for same-box A/B
# runs with tools/ab_sweep.py (diagnostic):  bash tools/make_ab_tree.sh treeB "-DDCSG_C4_WAVES=2"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=$2
T=$ROOT/tools/_build/$NAME
rm -rf "$T"; mkdir -p "$T/tests/golden"
cp -r "$ROOT/designcsg_amd" "$T/designcsg_amd"; rm -rf "$T/designcsg_amd/_build" "$T/designcsg_amd/__pycache__"
cp "$ROOT/bench.py" "$T/"; cp -r "$ROOT/include" "$T/include"
cp -r "$ROOT/tests/golden/scenes" "$T/tests/golden/scenes"
cd "$T" && DCSG_BUILD_DEFS="$DEFS" python3 -c "from designcsg_amd import build as b; b.build(verbose=False)"
echo "built $T with [$DEFS]"
