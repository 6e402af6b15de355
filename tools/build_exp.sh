#!/bin/bash
# tools/build_exp.sh NAME "-DFLAG ..." : a timing-experiment build of the engine (gpurun_exp/lib_NAME.so), checked with the
# same ISA lint and handler-layout check as the product build.  Used through PHYLY_AMD_LIB; never shipped.
set -e
name="$1"; flags="$2"
root="$(cd "$(dirname "$0")/.." && pwd)"; src="$root/phyly_amd/csrc"; out="/tmp/plk_exp_$name"
mkdir -p "$out" "$root/gpurun_exp"
cd "$src"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=on -I../../include -I. $flags --save-temps=obj -c plk_engine.hip -o "$out/plk_engine.o" 2> "$out/cc.log"
python3 "$root/tools/isa_lint.py" "$out/plk_engine-hip-amdgcn-amd-amdhsa-gfx950.s" | tail -1
[ -n "$SKIP_LAYOUT" ] || python3 "$root/tools/asm_layout_check.py" "$out/plk_engine-hip-amdgcn-amd-amdhsa-gfx950.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/gpurun_exp/lib_$name.so" "$out/plk_engine.o" host_*.o -lm -lpthread $(gcc -print-file-name=libquadmath.so)
echo "built gpurun_exp/lib_$name.so"
