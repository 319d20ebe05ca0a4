#!/bin/bash
# usage: bash tools/build_new.sh [keep]   -- lib/librovmpc.so -> librovmpc_prev.so (unless "keep"), rebuild, copy to librovmpc_new.so
cd "$(dirname "$0")/.." || exit 1
P=$(ls -d catenary-*_amd)
[ "$1" = "keep" ] || cp $P/lib/librovmpc.so $P/lib/librovmpc_prev.so
python3 -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i "error" -A5 | head -20
cp $P/lib/librovmpc.so $P/lib/librovmpc_new.so
ls -la $P/lib/librovmpc_new.so $P/lib/librovmpc_prev.so
