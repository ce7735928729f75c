#!/bin/bash
# build_variants/<name>/libfbk.so with extra -D flags (A/B runs of kernel variants: FBK_LIB_PATH selects the library)
#   scripts/build_variant.sh xcd -DFBK_V_XCD
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build_variants/$name
cd $R/featurebase_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" fbk.hip -o $R/build_variants/$name/libfbk.so
echo "built $name: $@"
