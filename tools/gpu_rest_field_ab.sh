#!/bin/bash
# tools/gpu_rest_field_ab.sh "name=flags" ...  (tools/gpu_rest_field_ab.py under the product and under each DNP_LIB variant)
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "## product"; python tools/gpu_rest_field_ab.py 2>&1 | grep -v amdgpu.ids
for V in "$@"; do
  NAME=${V%%=*}; FLAGS=${V#*=}
  python - <<PY
import sys
sys.path.insert(0, "$R")
from dipole_normal_prop_amd import build
build.build(extra_flags="$FLAGS".split(), out="$R/tools/bin/libdnp_$NAME.so", verbose=False)
PY
  echo "## $NAME: $FLAGS"
  DNP_LIB=$R/tools/bin/libdnp_$NAME.so python tools/gpu_rest_field_ab.py 2>&1 | grep -v amdgpu.ids
done
