#!/bin/bash
# A/B builds of libsvils.so: tools/build_variant.sh NAME -DFLAG...  -> svinet_amd/lib/libsvils_NAME.so  (every .hip of csrc/, one
# object per translation unit under svinet_amd/lib/obj_libsvils_NAME/, like the product build: svinet_amd/build.py)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python - "$name" "$@" <<'PY'
import sys
from svinet_amd import build
build._build_svils_variant("libsvils_" + sys.argv[1], list(sys.argv[2:]))
PY
