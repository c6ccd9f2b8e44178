#!/bin/bash
# A/B builds of libsvils.so: tools/build_variant.sh NAME -DFLAG...  -> svinet_amd/lib/libsvils_NAME.so
set -e
cd "$(dirname "$0")/../svinet_amd/csrc"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o ../lib/libsvils_$name.so svils_api.hip svils_device.hip svils_lpl.hip svils_report.hip
