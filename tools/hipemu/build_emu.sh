#!/bin/bash
# DEVELOPER-ONLY: compile libpebblegpu's sources with g++ against the fiber emulator (tools/hipemu).
# Output goes to tools/hipemu/_build/ and is loaded only by tools/hipemu/check.py.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
mkdir -p "$here/_build"
srcs=""
for f in "$root"/pebblesdr_amd/csrc/*.hip; do srcs="$srcs -x c++ $f"; done
g++ -O1 -g -std=c++17 -fPIC -shared -I"$here/shim" -I"$root/pebblesdr_amd/csrc" -Wno-unknown-pragmas \
    $srcs -x c++ "$root/pebblesdr_amd/csrc/design.cpp" -o "$here/_build/libpebblegpu_emu.so" -lm
echo "built $here/_build/libpebblegpu_emu.so"
